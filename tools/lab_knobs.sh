cd $GRAFT_REPO_ROOT
for cfg in "FRLW_WGRAD_PREC=0" "FRLW_WGRAD_TARGET=1280" "FRLW_WGRAD_TARGET=768" "FRLW_WGRAD_TARGET=512" "FRLW_WGRAD_TARGET=2048"; do
  echo "== $cfg"
  env $cfg timeout 120 build/wgrad_lab 64 10
done
