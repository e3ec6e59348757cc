cd $GRAFT_REPO_ROOT
for cfg in "72 1" "64 0"; do set -- $cfg
  for o in 0 1 default; do
    echo "== nlay $1 aerosols $2 overlap $o"
    if [ $o = default ]; then unset RRTMGP_HIP_STEP_OVERLAP; else export RRTMGP_HIP_STEP_OVERLAP=$o; fi
    NLAY=$1 AEROSOLS=$2 NCOLS=512,640,768,1024,1280,1792,2048,2560,4096,8192 python tools/experiments/small_step_host_cost.py 2>&1 | grep "^ncol" | cut -c1-36
  done
done
