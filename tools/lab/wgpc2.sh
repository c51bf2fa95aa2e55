# Two half-size workgroups per CU in the fused tile kernels: A/B through the C ABI.
#   build (where hipcc runs):  python tools/lab/wgpc2.py --build-only
#   on the GPU box:            bash tools/lab/wgpc2.sh
set -x
L=gconv-adapter_amd/lib
echo "== product build (one 10 / 8-wave workgroup per CU, T = N / 256)"
python tools/tilebench.py hiv esol hiv4096
echo "== product build, T = N / 512 (more, smaller tiles; still one workgroup per CU)"
GCONV_TILE_NODES_OVERRIDE=26 GCONV_TILE_SLACK_DIV=4 python tools/tilebench.py hiv
echo "== WGPC=2 build (5-wave workgroups, <= 168 VGPRs, grid 2 x CUs), T = N / 256 and N / 512"
GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so python tools/tilebench.py hiv esol hiv4096
for T in 20 26 32 40; do
  GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so GCONV_TILE_NODES_OVERRIDE=$T GCONV_TILE_SLACK_DIV=4 python tools/tilebench.py hiv
done
GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so GCONV_TILE_NODES_OVERRIDE=50 GCONV_TILE_SLACK_DIV=2 python tools/tilebench.py hiv4096
echo "== where the workgroups ran (stamp build of WGPC=2, T = 26): co-residency per CU"
GCONV_STAMPS_LIB=$L/libgconv_adapter_wgpc2_stamps.so GCONV_TILE_NODES_OVERRIDE=26 GCONV_TILE_SLACK_DIV=4 GCONV_STAMPS_GRID=512 python tools/tilestamps.py hiv
L=gconv-adapter_amd/lib
export GCONV_BENCH_HIDDEN=256
echo "== d=256 product"; python tools/tilebench.py hiv hiv4096
echo "== d=256 product T=26"; GCONV_TILE_NODES_OVERRIDE=26 GCONV_TILE_SLACK_DIV=4 python tools/tilebench.py hiv
echo "== d=256 WGPC2"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so python tools/tilebench.py hiv hiv4096
for T in 26 32 40; do echo "== d=256 WGPC2 T=$T"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so GCONV_TILE_NODES_OVERRIDE=$T GCONV_TILE_SLACK_DIV=4 python tools/tilebench.py hiv; done
echo "== d=256 WGPC2 hiv4096 T=50"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so GCONV_TILE_NODES_OVERRIDE=50 GCONV_TILE_SLACK_DIV=2 python tools/tilebench.py hiv4096
GCONV_STAMPS_LIB=$L/libgconv_adapter_wgpc2_stamps.so GCONV_TILE_NODES_OVERRIDE=26 GCONV_TILE_SLACK_DIV=4 GCONV_STAMPS_GRID=512 python tools/tilestamps.py hiv 2>&1 | cut -c1-400
