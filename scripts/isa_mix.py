#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels: python scripts/isa_mix.py file.s [kernel-substring]
(file.s from `hipcc --cuda-device-only -S`).  Counts opcodes per kernel; no GPU needed."""
import collections, re, sys
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None; mix = collections.OrderedDict()
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur = m.group(1); mix[cur] = collections.Counter(); continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"): cur = None; continue
    if cur is None: continue
    m = re.match(r"^\t([a-z_0-9]+)\s", line)
    if m and not m.group(1).startswith("."): mix[cur][m.group(1)] += 1
for k, c in mix.items():
    if want not in k or not c: continue
    tot = sum(c.values())
    grp = collections.Counter()
    for op, n in c.items():
        if op.startswith("v_") and "f64" in op and "mfma" not in op: grp["valu_f64"] += n
        elif "mfma" in op: grp["mfma"] += n
        elif op.startswith("v_accvgpr"): grp["accvgpr_mov"] += n
        elif op.startswith("v_"): grp["valu_other"] += n
        elif op.startswith("s_"): grp["salu"] += n
        elif op.startswith("ds_"): grp["lds"] += n
        elif op.startswith("global_atomic"): grp["global_atomic"] += n
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): grp["vmem"] += n
        elif op.startswith("scratch_"): grp["scratch"] += n
        else: grp["other"] += n
    print("== %s: %d instructions" % (k, tot))
    print("   groups:", dict(grp))
    print("   top:", ", ".join("%s %d" % x for x in c.most_common(28)))
