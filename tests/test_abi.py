"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/kid.h declares, and agrees with
the header about struct layouts.  No compute calls (there is no GPU in the build container)."""
import ctypes as C
import os
import re

from icebergs_amd import lib as L
from icebergs_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    L.build()
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "kid.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kid_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match():
    lib = L.load()
    assert lib.kid_sizeof(0) == C.sizeof(T.Params)
    assert lib.kid_sizeof(1) == C.sizeof(T.GridDesc)
    assert lib.kid_sizeof(2) == C.sizeof(T.BergSoA)
    assert lib.kid_sizeof(3) == C.sizeof(T.BondSoA)
    assert lib.kid_sizeof(4) == C.sizeof(T.ForcingIn)
    assert lib.kid_sizeof(5) == C.sizeof(T.CalvingParams) and lib.kid_sizeof(6) == C.sizeof(T.CalvingIn)
    assert lib.kid_version().startswith(b"kid_hip")
    assert b"EXPERIMENTS" not in lib.kid_version()   # the product library is not a measurement build (KID_EXP_* macros)


def test_no_gpu_means_loud_failure():
    """Without a device kid_create must fail (KID_ENODEV), never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    from icebergs_amd import synthetic as S
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c1()
    try:
        Icebergs(grid, p, capacity=16)
    except L.KidError as e:
        assert "rc=-3" in str(e) or "rc=-2" in str(e)
    else:
        raise AssertionError("kid_create succeeded without a GPU")


def test_product_never_touches_the_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "icebergs_amd")
    banned = re.compile(r"oracle_lib|kid_oracle|libkid_oracle|oracle/|import\s+oracle|from\s+oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".F90", ".f90", ".c", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not banned.search(text), os.path.join(dirpath, f)


def test_structs_have_no_implicit_padding():
    """Fortran stream I/O and bind(C) derived types move components one by one: every pad must be spelled out"""
    import ctypes
    from icebergs_amd import types as T
    for cls in (T.Params, T.GridDesc, T.BergSoA, T.BondSoA, T.ForcingIn, T.CalvingParams, T.CalvingIn, T.TrajParams):
        assert ctypes.sizeof(cls) == sum(ctypes.sizeof(t) for _, t in cls._fields_), cls.__name__


def test_fortran_module_covers_the_header():
    """Every entry point of include/kid.h has a bind(C) interface in the Fortran module (the host language north_star names) and
    is public there, and every type of include/kid_types.h that crosses the boundary is exported."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "kid.h")).read()
    mod = open(os.path.join(root, "icebergs_amd", "fortran", "kid_hip_mod.F90")).read()
    inc = open(os.path.join(root, "icebergs_amd", "fortran", "kid_types_gen.inc")).read()
    entry = set(re.findall(r"\b(kid_[a-z0-9_]+)\s*\(", hdr))
    binds = set(re.findall(r"bind\(C, name='(kid_[a-z0-9_]+)'\)", mod))
    public = set()
    for m in re.finditer(r"^\s*public ::(.*)$", mod, re.M):
        public |= {x.strip() for x in m.group(1).split(",")}
    types = set(re.findall(r"type, bind\(C\) :: (\w+)", inc))
    assert not (entry - binds - types), sorted(entry - binds - types)     # (a type's name followed by '(' is a cast in a comment, not a function)
    assert not (binds - public), sorted(binds - public)
    assert not (types - public), sorted(types - public)


def test_icebergs_nml_table_matches_the_reference_namelist():
    """tools/icebergs_nml_table.py (the declarations the Fortran glue reads &icebergs_nml with) lists exactly the variables of the
    reference's namelist statement, in its order (icebergs_framework.F90:823-856).  Runs where the reference tree is present (the
    build container); a Fortran namelist read fails on any name the group does not declare, so a missing one would break a
    maintainer's input.nml."""
    import os
    import re
    import sys
    ref = "/root/reference/src/icebergs_framework.F90"
    if not os.path.exists(ref):
        import pytest
        pytest.skip("reference tree not present")
    src = open(ref).read()
    m = re.search(r"namelist\s*/icebergs_nml/(.*?)\n\s*\n", src, re.S | re.I)
    assert m, "namelist statement not found"
    body = re.sub(r"!.*", "", m.group(1))
    body = body.replace("&", " ")
    names = [x.strip() for x in body.replace("\n", " ").split(",") if x.strip()]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from icebergs_nml_table import ICEBERGS_NML
    mine = [r[0] for r in ICEBERGS_NML]
    assert [n.lower() for n in names] == [n.lower() for n in mine]
    assert len(mine) == 163
    inc = open(os.path.join(root, "icebergs_amd", "fortran", "kid_nml_gen.inc")).read().lower()
    for n in mine:
        assert re.search(r"::\s*%s\b" % re.escape(n.lower()), inc), n      # declared in the generated include


def test_generated_fortran_namelist_includes_are_current():
    """kid_nml_gen.inc / kid_nml_defaults_gen.inc are what tools/gen_fortran_nml.py makes of the table today"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = [os.path.join(root, "icebergs_amd", "fortran", f) for f in ("kid_nml_gen.inc", "kid_nml_defaults_gen.inc")]
    before = [open(p).read() for p in paths]
    try:
        subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fortran_nml.py")], check=True, capture_output=True)
        assert [open(p).read() for p in paths] == before
    finally:
        for p, text in zip(paths, before):
            open(p, "w").write(text)
