cd $GRAFT_REPO_ROOT
for cfg in "FRLW_CONV_RING4=0" "FRLW_CONV_RING3=1" "FRLW_CONV_RING4=1" "FRLW_CONV_RING4=1 FRLW_CONV_SPLIT_BELOW=0"; do
  echo "== $cfg"
  env $cfg timeout 120 build/conv_lab 32 20 1000 1 | grep -v "^sum"
done
