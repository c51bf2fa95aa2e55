"""Per-kernel ISA statistics from a hipcc -S dump: python tools/isa_stats.py file.s [name-substr ...]"""
import re, sys
text = open(sys.argv[1]).read()
subs = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if subs and not any(s in name for s in subs):
        continue
    c = lambda pat: len(re.findall(pat, body))
    vg = re.search(r"\.name:\s+%s\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)" % re.escape(name), text)
    v0, l0 = c(r"vmcnt\(0\)"), c(r"lgkmcnt\(0\)")
    print(f"{name[:70]:70s} lines={body.count(chr(10)):5d} mfma={c('v_mfma'):4d} gld4={c('global_load_dwordx4'):3d} "
          f"gld1={c('global_load_dword '):3d} gst4={c('global_store_dwordx4'):3d} vmcnt0={v0:3d} "
          f"lgkm0={l0:3d} scratch={c('scratch_'):3d} vgpr={vg.group(1) if vg else '?'}")
