# A-fragment read distance in the W-stationary GEMM (GCONV_MLP_WS_SCHED): product (2) against 0 / 1 / 3
cd $GRAFT_REPO_ROOT
L=gconv-adapter_amd/lib
python -m pytest tests/test_gin.py -m gpu -x -q 2>&1 | tail -2
echo "== product (2 ahead)"; python tools/lab/ws_gemm.py
for v in ahead0 ahead1 ahead3; do
  echo "== $v"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_ws_$v.so python tools/lab/ws_gemm.py
done
echo "== product again"; python tools/lab/ws_gemm.py
