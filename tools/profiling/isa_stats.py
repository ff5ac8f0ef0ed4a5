#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing, split at the `; KIDMARK name` markers that
-DKID_EXP_MARKERS leaves behind:  isa_stats.py file.s [kernel-substring]"""
import collections, re, sys
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "berg_kernel"
lines = open(path).read().split("\n")
inside = False; seg = "start"; ops = collections.OrderedDict()
for l in lines:
    if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l): inside = True; seg = "start"; continue
    if inside and l.startswith(".Lfunc_end"): break
    if not inside: continue
    m = re.search(r"; KIDMARK (\S+)", l)
    if m: seg = m.group(1); continue
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"): continue
    ops.setdefault(seg, collections.Counter())[t.split()[0]] += 1
tot = collections.Counter()
def row(name, o):
    c = sum(o.values())
    f = lambda pred: sum(v for k, v in o.items() if pred(k))
    print("%-20s total %5d valu %5d f64 %5d mov %4d cnd %4d lane %4d salu %5d ds %4d mem %4d div %3d rcp %3d rsq %3d call %3d nop %3d" % (
        name, c, f(lambda k: k.startswith("v_")), f(lambda k: "f64" in k), f(lambda k: k.startswith("v_mov")), f(lambda k: k.startswith("v_cndmask")),
        f(lambda k: "readlane" in k or "writelane" in k), f(lambda k: k.startswith("s_")), f(lambda k: k.startswith("ds_")),
        f(lambda k: k.startswith(("global_", "flat_", "scratch_", "buffer_"))), o.get("v_div_fmas_f64", 0), f(lambda k: k.startswith("v_rcp_f64")),
        f(lambda k: k.startswith("v_rsq_f64")), o.get("s_swappc_b64", 0), o.get("s_nop", 0)))
for s, o in ops.items():
    row(s, o); tot.update(o)
row("ALL", tot)
