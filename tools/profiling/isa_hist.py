#!/usr/bin/env python3
"""Opcode histogram of one kernel in a hipcc -S listing (static counts), per KIDMARK segment and overall:
   isa_hist.py file.s [kernel-substring] [top-N]"""
import collections, re, sys
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "berg_kernel"; top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
inside = False; seg = "start"; ops = collections.OrderedDict()
for l in open(path).read().split("\n"):
    if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l): inside = True; seg = "start"; continue
    if inside and l.startswith(".Lfunc_end"): break
    if not inside: continue
    m = re.search(r"; KIDMARK (\S+)", l)
    if m: seg = m.group(1); continue
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"): continue
    ops.setdefault(seg, collections.Counter())[t.split()[0]] += 1
tot = collections.Counter()
for s, o in ops.items(): tot.update(o)
def cls(k):
    if not k.startswith("v_"): return "salu" if k.startswith("s_") else ("lds" if k.startswith("ds_") else "vmem/other")
    if k.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")): return "valu fp64 add/mul/fma"
    if k.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")): return "valu transcendental"
    if k.startswith("v_cmp") or k.startswith("v_cmpx"): return "valu compare"
    if k.startswith("v_cndmask"): return "valu select (v_cndmask_b32)"
    if k.startswith(("v_mov", "v_accvgpr", "v_pk_mov")): return "valu move"
    if k.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "valu lane<->scalar"
    if "f64" in k: return "valu other fp64 (min/max/ldexp/frexp/cvt/fract/rndne/div_*)"
    return "valu integer / fp32 / bit"
def show(name, o):
    n = sum(o.values()); c = collections.Counter()
    for k, v in o.items(): c[cls(k)] += v
    print("== %s: %d instructions" % (name, n))
    for k, v in c.most_common(): print("   %-62s %5d  %4.1f %%" % (k, v, 100.0 * v / n))
    print("   top opcodes: " + ", ".join("%s %d" % kv for kv in o.most_common(top)))
show("whole kernel (static)", tot)
for s, o in ops.items(): show("segment after marker '%s'" % s, o)
