# W-stationary f16x2 GEMM (csrc/mlp_f16.hip, GCONV_MLP_WS) against the tiled kernel: the whole GPU suite first,
# then the MLP / frozen-layer / fine-tune-step timings, each in its own process (the knob is read once).
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
for ws in 1 0; do
  echo "== GCONV_MLP_WS=$ws"
  GCONV_MLP_WS=$ws python tools/lab/mlp_bench.py 2>&1 | tail -2
  GCONV_MLP_WS=$ws python tools/lab/mlp_bench.py 102952 2>&1 | tail -2
  GCONV_MLP_WS=$ws python tools/finetunestep.py 100 2>&1 | tail -1
done
