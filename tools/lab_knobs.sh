cd $GRAFT_REPO_ROOT
for cfg in "FRLW_CONV_ROW4_MIN=100000" "FRLW_CONV_ROW4_MIN=600" "FRLW_CONV_ROW4_MIN=100" "FRLW_CONV_ROW4_MIN=1"; do
  echo "== $cfg"
  env $cfg timeout 120 build/conv_lab 32 20 1000 1 | grep -v "^sum"
done
