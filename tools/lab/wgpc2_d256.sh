# d = 256 (4 chunks -> 4-wave workgroups in the WGPC = 2 build: one wave per SIMD, so two workgroups CAN share a CU)
L=gconv-adapter_amd/lib
export GCONV_BENCH_HIDDEN=256
echo "== d=256 product"; python tools/tilebench.py hiv hiv4096
echo "== d=256 product T=26"; GCONV_TILE_NODES_OVERRIDE=26 GCONV_TILE_SLACK_DIV=4 python tools/tilebench.py hiv
echo "== d=256 WGPC2"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so python tools/tilebench.py hiv hiv4096
for T in 26 32 40; do echo "== d=256 WGPC2 T=$T"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so GCONV_TILE_NODES_OVERRIDE=$T GCONV_TILE_SLACK_DIV=4 python tools/tilebench.py hiv; done
echo "== d=256 WGPC2 hiv4096 T=50"; GCONV_ADAPTER_LIB=$L/libgconv_adapter_wgpc2.so GCONV_TILE_NODES_OVERRIDE=50 GCONV_TILE_SLACK_DIV=2 python tools/tilebench.py hiv4096
GCONV_STAMPS_LIB=$L/libgconv_adapter_wgpc2_stamps.so GCONV_TILE_NODES_OVERRIDE=26 GCONV_TILE_SLACK_DIV=4 GCONV_STAMPS_GRID=512 python tools/tilestamps.py hiv 2>&1 | cut -c1-400
