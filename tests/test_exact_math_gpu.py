"""The -DKID_EXACT_MATH twin of the library (icebergs_amd/csrc/libkid_hip_exact.so, built by build()) against the oracle, bit by bit.

The default build trims the hot loop's arithmetic (Newton-refined reciprocals and roots, the fifth-root form of the melt laws,
series for the RK4 stages' sin/cos, explicit fma, unrotated cells passing through): its results differ from the oracle's at the
1e-16 level, inside the 1e-10 tolerance of the parity tests.  This file is the check that those trims -- and only those -- are
the difference: with every one of them switched back to the IEEE operation the oracle performs, what is left between the two
sides is the mathematical library (ocml's sin, cos and pow on the device, glibc's on the host; both within 1 ulp, not the same
bits), and results must be EQUAL wherever no such function is involved:

  * config 1 (Cartesian grid, f-plane: no trigonometric function on the path of a berg) -- lon, lat, uvel, vvel, axn .. yj bit
    for bit over the 144 steps, RK4 and Verlet; sizes and masses (the melt laws use pow) within a few ulp;
  * config 2 (lat-lon grid: sincos of the latitude in every step) -- trajectories within 1e-13 relative, three orders below the
    parity tolerance; nothing may hide behind it.
The test runs in a child process so that the twin is the only copy of the library in it (KID_HIP_SO)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = os.path.join(ROOT, "icebergs_amd", "csrc", "libkid_hip_exact.so")

CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/oracle"); sys.path.insert(0, %(root)r + "/tests")
from icebergs_amd import synthetic as S, lib
import parity as P
assert b"exact-math" in lib.load().kid_version(), lib.load().kid_version()
def ulps(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    ia, ib = a.view(np.int64), b.view(np.int64)
    ia = np.where(ia < 0, np.int64(-2**63) - ia, ia); ib = np.where(ib < 0, np.int64(-2**63) - ib, ib)
    return int(np.abs(ia - ib).max()) if a.size else 0
out = {}
for name, build, nsteps in (("c1_rk4", lambda: S.config_c1(), 144),
                            ("c1_verlet", lambda: (lambda g, p, b: (g, (setattr(p, "Runge_not_Verlet", 0), setattr(p, "old_bug_bilin", 0), p)[2], b))(*S.config_c1()), 144),
                            ("c2", lambda: S.config_c2(n=40000, seed=2), 48)):
    grid, p, b = build()
    ref = P.run_oracle(grid, p, b, nsteps)
    got = P.run_hip(grid, p, b, nsteps, mode="fused")
    rb, gb = ref[0], got[0]
    ra, ga = np.nonzero(rb["alive"] != 0)[0], np.nonzero(gb["alive"] != 0)[0]   # survivors (the library drops dead rows when it re-bins)
    o1, o2 = ra[np.argsort(rb["id"][ra])], ga[np.argsort(gb["id"][ga])]
    assert np.array_equal(rb["id"][o1], gb["id"][o2]), "survivors differ"
    res = {}
    for f in P.TRAJ_FIELDS + P.SIZE_FIELDS:
        res[f] = {"ulps": ulps(gb[f][o2], rb[f][o1]), "rel": P.rel_err(gb[f][o2], rb[f][o1])}
    res["cells_equal"] = bool(np.array_equal(rb["ine"][o1], gb["ine"][o2]) and np.array_equal(rb["jne"][o1], gb["jne"][o2]))
    out[name] = res
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_exact_math_build_equals_the_oracle(oracle):
    assert os.path.exists(EXACT), "libkid_hip_exact.so is not built (build() makes it)"
    env = dict(os.environ, KID_HIP_SO=EXACT)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    import parity as P
    for case in ("c1_rk4", "c1_verlet"):
        res = out[case]
        assert res["cells_equal"], case
        for f in P.TRAJ_FIELDS:   # no libm on this path: the same IEEE operations in the same order
            assert res[f]["ulps"] == 0, (case, f, res[f])
        for f in P.SIZE_FIELDS:   # the melt laws: pow (ocml vs glibc)
            assert res[f]["rel"] <= 1e-14, (case, f, res[f])
    res = out["c2"]
    assert res["cells_equal"]
    for f in P.TRAJ_FIELDS + P.SIZE_FIELDS:   # sincos(lat) every step, pow: libm only
        assert res[f]["rel"] <= 1e-13, ("c2", f, res[f])
    print(json.dumps({c: {f: v["ulps"] for f, v in r_.items() if isinstance(v, dict)} for c, r_ in out.items()}))
