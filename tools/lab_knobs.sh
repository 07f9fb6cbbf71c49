cd $GRAFT_REPO_ROOT
for cfg in "FRLW_CONV_W2=0"; do
  echo "== $cfg"
  env $cfg timeout 120 build/conv_lab 32 20 1000 1 | grep -v "^sum"
done
