"""ctypes mirrors of include/kid_types.h.

The header is the single source of truth: enums and the ``kid_params`` / ``kid_grid_desc`` /
``kid_berg_soa`` structs are parsed from it at import time, so the Python host, the HIP library and
the CPU oracle can never disagree about a field offset.  (Reference types these flatten:
icebergs_framework.F90:112-229 icebergs_gridded, :290-359 iceberg, :421-616 icebergs.)
"""
import ctypes as C
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "kid_types.h")


def _strip_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def _parse_enums(src):
    names = {}
    for body in re.findall(r"enum\s*\{(.*?)\}\s*;", src, flags=re.S):
        val = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [t.strip() for t in item.split("=", 1)]
                val = int(eval(expr, {"__builtins__": {}}, names))  # noqa: S307 - header-only arithmetic
            else:
                name, val = item, val + 1
            names[name] = val
    return names


def _parse_struct(src, name, consts):
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), src, flags=re.S)
    fields = []
    ctype = {"double": C.c_double, "int32_t": C.c_int32, "int64_t": C.c_int64}
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^const\s+", "", decl)
        mm = re.match(r"(double|int32_t|int64_t)\s*(\*?)\s*(.*)", decl)
        base, ptr, rest = mm.group(1), mm.group(2), mm.group(3)
        for var in rest.split(","):
            var = var.strip()
            isptr = bool(ptr) or var.startswith("*")
            var = var.lstrip("* ")
            arr = re.match(r"(\w+)\[(\w+)\]", var)
            t = C.POINTER(ctype[base]) if isptr else ctype[base]
            if arr:
                n = arr.group(2)
                n = int(n) if n.isdigit() else consts[n]
                fields.append((arr.group(1), t * n))
            else:
                fields.append((var, t))
    return fields


_src = _strip_comments(open(_HDR).read())
ENUMS = _parse_enums(_src)
globals().update(ENUMS)


class GridDesc(C.Structure):
    _fields_ = _parse_struct(_src, "kid_grid_desc", ENUMS)


class Params(C.Structure):
    _fields_ = _parse_struct(_src, "kid_params", ENUMS)


class BergSoA(C.Structure):
    _fields_ = _parse_struct(_src, "kid_berg_soa", ENUMS)


class ForcingIn(C.Structure):
    _fields_ = _parse_struct(_src, "kid_forcing_in", ENUMS)


class CalvingParams(C.Structure):
    _fields_ = _parse_struct(_src, "kid_calving_params", ENUMS)


class TrajParams(C.Structure):
    _fields_ = _parse_struct(_src, "kid_traj_params", ENUMS)


class CalvingIn(C.Structure):
    _fields_ = _parse_struct(_src, "kid_calving_in", ENUMS)


class BondSoA(C.Structure):
    _fields_ = _parse_struct(_src, "kid_bond_soa", ENUMS)


BOND_F64_NAMES = [k[len("KID_BOND_"):].lower() for k, v in sorted(
    ((k, v) for k, v in ENUMS.items() if k.startswith("KID_BOND_")), key=lambda kv: kv[1])]
BERG_F64_NAMES = [k[len("KID_B_"):].lower() for k, v in sorted(
    ((k, v) for k, v in ENUMS.items() if k.startswith("KID_B_")), key=lambda kv: kv[1])]
BERG_I32_NAMES = [k[len("KID_BI_"):].lower() for k, v in sorted(
    ((k, v) for k, v in ENUMS.items() if k.startswith("KID_BI_")), key=lambda kv: kv[1])]
GRID_STATIC_NAMES = [k[len("KID_G_"):].lower() for k, v in sorted(
    ((k, v) for k, v in ENUMS.items() if k.startswith("KID_G_")), key=lambda kv: kv[1])]
FORCING_NAMES = [k[len("KID_F_"):].lower() for k, v in sorted(
    ((k, v) for k, v in ENUMS.items() if k.startswith("KID_F_") and not k.startswith("KID_FL_")), key=lambda kv: kv[1])]
ACC_NAMES = {k[len("KID_A_"):].lower(): v for k, v in ENUMS.items() if k.startswith("KID_A_")}
OUT_NAMES = {k[len("KID_O_"):].lower(): v for k, v in ENUMS.items() if k.startswith("KID_O_")}
SCALAR_NAMES = {k[len("KID_S_"):].lower(): v for k, v in ENUMS.items() if k.startswith("KID_S_")}
NACC = ENUMS["KID_NACC"]
NOUT = ENUMS["KID_NOUT"]
NSCALAR = ENUMS["KID_NSCALAR"]
