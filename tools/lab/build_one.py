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
obj = os.path.join(objdir, f"{stem}_lab_{name}.o")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value", "-Wno-pass-failed",
                *defines, "-o", obj, os.path.join(g.PKG, "csrc", stem + ".hip")], check=True, cwd=g.PKG)
sources = sorted(glob.glob(os.path.join(g.PKG, "csrc", "*.hip")))
objs, _ = g._compile_objects(sources, False, (), "")
objs = [obj if os.path.basename(o) == stem + ".o" else o for o in objs]
out = os.path.join(g.PKG, "lib", f"libgconv_adapter_{name}.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=g.PKG)
print(out)
