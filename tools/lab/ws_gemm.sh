# leave-one-out timings of the W-stationary GEMM (tools/lab/ws_gemm.py --build first)
cd $GRAFT_REPO_ROOT
L=gconv-adapter_amd/lib
python -m pytest tests/test_gin.py -m gpu -x -q 2>&1 | tail -3
echo "== product"; python tools/lab/ws_gemm.py
echo "== tiled (GCONV_MLP_WS=0)"; GCONV_MLP_WS=0 python tools/lab/ws_gemm.py
for v in noepi nostore nodma nomfma_noepi; do
  echo "== $v"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_ws_$v.so python tools/lab/ws_gemm.py
done
