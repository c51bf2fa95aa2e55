"""Compact control-flow / memory-op view of one kernel: python tools/isa_flow.py file.s name-substr [from_line]"""
import re, sys
text = open(sys.argv[1]).read()
sub = sys.argv[2]
start = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    if sub not in m.group(1):
        continue
    body = m.group(2).split("\n")
    run = 0
    for i, l in enumerate(body):
        t = l.strip()
        if not re.search(r"global_load|global_store|vmcnt|lgkmcnt|s_cbranch|s_branch|v_mfma|s_barrier|ds_read|ds_write|^\.LBB", t):
            continue
        if "v_mfma" in t:
            run += 1
            continue
        if run:
            print(f"        [{run} mfma]")
            run = 0
        if i >= start:
            print(f"{i:5d} {t[:110]}")
    break
