"""Lab library that differs from the product in ONE source: compiles `csrc/<stem>.hip` with the given defines and links
it with the product's other objects (a full `build_library(defines=...)` recompiles every source: 2 minutes).
  usage: python3 tools/lab/build_one.py <stem> <name> -DX=1 [-DY=2 ...]   ->  gconv-adapter_amd/lib/libgconv_adapter_<name>.so"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

stem, name, defines = sys.argv[1], sys.argv[2], sys.argv[3:]
g.build_library()   # the product objects, up to date
objdir = os.path.join(g.PKG, "build")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sources = sorted(glob.glob(os.path.join(g.PKG, "csrc", "*.hip")))
objs, _ = g._compile_objects(sources, False, (), "")
# (fused_tile is compiled once per padded rank: fused_tile_fp4.o ... fused_tile_host.o, each with its own -D)
mine = [o for o in objs if os.path.basename(o) == stem + ".o" or os.path.basename(o).startswith(stem + "_")]
from concurrent.futures import ThreadPoolExecutor
jobs = []
for o in mine:
    suffix = os.path.basename(o)[len(stem):-2]   # "", "_fp8", "_host"
    extra = []
    if suffix.startswith("_fp"):
        extra = [f"-DGCONV_TILE_ONLY_FP={suffix[3:]}"]
    elif suffix == "_host":
        extra = ["-DGCONV_TILE_ONLY_FP=0"]
    lab = os.path.join(objdir, f"{stem}{suffix}_lab_{name}.o")
    jobs.append((o, lab, [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value",
                          "-Wno-pass-failed", *defines, *extra, "-o", lab, os.path.join(g.PKG, "csrc", stem + ".hip")]))
with ThreadPoolExecutor(len(jobs)) as ex:
    for r in ex.map(lambda j: subprocess.run(j[2], cwd=g.PKG), jobs):
        if r.returncode != 0:
            raise SystemExit(r.returncode)
swap = {j[0]: j[1] for j in jobs}
objs = [swap.get(o, o) for o in objs]
out = os.path.join(g.PKG, "lib", f"libgconv_adapter_{name}.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=g.PKG)
print(out)
