#!/usr/bin/env python3
"""Inner loops of one kernel in a gfx950 .s file (hipcc --save-temps): for every backward branch, the instruction mix
between its target label and the branch.   tools/isa_loops.py file.s kernel_substring [min_instructions]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def kind(op):
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith(("v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64", "v_cvt_f32_f64", "v_cvt_i32_f64")): return "v_f64"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "vmem_ld"
    if op.startswith("global_store") or op.startswith("scratch_"): return "vmem_st/scratch"
    if op.startswith("ds_"): return "lds"
    return "other"
print("kernel at line", start + 1, "length", len(body))
for i, l in enumerate(body):
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        cnt = {}
        for x in body[a:i + 1]:
            x = x.strip()
            if not x or x.startswith((";", ".")) or x.endswith(":"): continue
            k = kind(x.split()[0])
            cnt[k] = cnt.get(k, 0) + 1
        tot = sum(cnt.values())
        if tot >= minn:
            print("loop %s lines %d..%d: %d instructions  %s" % (m.group(1), start + a + 1, start + i + 1, tot, dict(sorted(cnt.items()))))
