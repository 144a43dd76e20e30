"""Reader for the CaNS/CaLES case file ``input.nml`` (host side of the drop-in boundary).

Mirrors ``read_input`` (reference src/param.f90:88-224): namelist groups ``&dns`` and
``&les`` are required, ``&cudecomp`` is parsed and ignored, ``dt_f`` defaults to -1
(param.f90:124) and ``dl, dli, visc`` are derived as in param.f90:153-157. A group may be
closed by ``\\`` instead of ``/`` -- every ``examples/dns/*/input.nml`` of the reference
does that for ``&les`` -- and is accepted here.

Array-valued entries keep the Fortran index order: ``cbcvel[side, dir, vel]``,
``cbcpre[side, dir]``, ``lwm[side, dir]`` with side in {0,1} and 0-based dir/vel.
"""
from __future__ import annotations

import dataclasses
import re
from typing import Any, Dict, List, Tuple

import numpy as np

# name -> (type, shape, lower bounds)   (declarations: src/param.f90:37-76)
_SCHEMA: Dict[str, Tuple[str, Tuple[int, ...], Tuple[int, ...]]] = {
    "ng": ("i", (3,), (1,)), "l": ("r", (3,), (1,)), "gtype": ("i", (), ()), "gr": ("r", (), ()),
    "cfl": ("r", (), ()), "dtmax": ("r", (), ()), "dt_f": ("r", (), ()), "visci": ("r", (), ()),
    "inivel": ("c", (), ()), "is_wallturb": ("l", (), ()), "nstep": ("i", (), ()),
    "time_max": ("r", (), ()), "tw_max": ("r", (), ()), "stop_type": ("l", (3,), (1,)),
    "restart": ("l", (), ()), "is_overwrite_save": ("l", (), ()), "nsaves_max": ("i", (), ()),
    "icheck": ("i", (), ()), "iout0d": ("i", (), ()), "iout1d": ("i", (), ()), "iout2d": ("i", (), ()),
    "iout3d": ("i", (), ()), "isave": ("i", (), ()),
    "cbcvel": ("c", (2, 3, 3), (0, 1, 1)), "cbcpre": ("c", (2, 3), (0, 1)), "cbcsgs": ("c", (2, 3), (0, 1)),
    "bcvel": ("r", (2, 3, 3), (0, 1, 1)), "bcpre": ("r", (2, 3), (0, 1)), "bcsgs": ("r", (2, 3), (0, 1)),
    "bforce": ("r", (3,), (1,)), "is_forced": ("l", (3,), (1,)), "velf": ("r", (3,), (1,)),
    "dims": ("i", (2,), (1,)),
    "sgstype": ("c", (), ()), "lwm": ("i", (2, 3), (0, 1)), "hwm": ("r", (), ()),
}
_GROUPS = {"dns": [k for k in _SCHEMA if k not in ("sgstype", "lwm", "hwm")], "les": ["sgstype", "lwm", "hwm"]}


class NamelistError(ValueError):
    """Raised for what the reference treats as a fatal input error (param.f90:126-152)."""


def _strip_comments(text: str) -> str:
    out = []
    for line in text.splitlines():
        q = None
        for i, ch in enumerate(line):
            if q:
                if ch == q:
                    q = None
            elif ch in "'\"":
                q = ch
            elif ch == "!":
                line = line[:i]
                break
        out.append(line)
    return "\n".join(out)


def _split_groups(text: str) -> Dict[str, str]:
    groups: Dict[str, str] = {}
    for m in re.finditer(r"&\s*(\w+)(.*?)(?:^\s*[/\\]\s*$|[/\\]\s*(?=\n\s*&|\s*\Z))", text, re.S | re.M):
        groups[m.group(1).lower()] = m.group(2)
    return groups


def _tokens(values: str) -> List[str]:
    toks = re.findall(r"'[^']*'|\"[^\"]*\"|[^\s,]+", values)
    out: List[str] = []
    for t in toks:
        m = re.fullmatch(r"(\d+)\*(.+)", t)
        if m:
            out.extend([m.group(2)] * int(m.group(1)))
        else:
            out.append(t)
    return out


def _convert(tok: str, typ: str) -> Any:
    if typ == "c":
        return tok[1:-1] if tok[:1] in "'\"" else tok
    if typ == "l":
        t = tok.strip(".").lower()
        if t[:1] == "t":
            return True
        if t[:1] == "f":
            return False
        raise NamelistError(f"bad logical value {tok!r}")
    if typ == "i":
        return int(tok)
    return float(tok.lower().replace("d", "e"))


def _section(spec: str | None, shape: Tuple[int, ...], lb: Tuple[int, ...]) -> List[Tuple[int, ...]]:
    """Element list (0-based index tuples) of a Fortran array section, first index fastest."""
    if not shape:
        return [()]
    ranges = []
    parts = [p.strip() for p in spec.strip("()").split(",")] if spec else [":"] * len(shape)
    if len(parts) != len(shape):
        raise NamelistError(f"rank mismatch in section {spec!r}")
    for p, n, l0 in zip(parts, shape, lb):
        if ":" in p:
            a, b = (p.split(":") + [""])[:2]
            lo = int(a) if a.strip() else l0
            hi = int(b) if b.strip() else l0 + n - 1
            ranges.append(range(lo - l0, hi - l0 + 1))
        else:
            ranges.append(range(int(p) - l0, int(p) - l0 + 1))
    out: List[Tuple[int, ...]] = []

    def rec(d: int, cur: Tuple[int, ...]) -> None:   # last dimension outermost = first index fastest
        if d < 0:
            out.append(cur)
            return
        for i in ranges[d]:
            rec(d - 1, (i,) + cur)

    rec(len(ranges) - 1, ())
    return out


@dataclasses.dataclass
class Case:
    """Parsed ``input.nml`` plus the derived quantities of param.f90:153-157."""
    ng: np.ndarray
    l: np.ndarray
    gtype: int
    gr: float
    cfl: float
    dtmax: float
    dt_f: float
    visci: float
    inivel: str
    is_wallturb: bool
    nstep: int
    time_max: float
    tw_max: float
    stop_type: np.ndarray
    restart: bool
    is_overwrite_save: bool
    nsaves_max: int
    icheck: int
    iout0d: int
    iout1d: int
    iout2d: int
    iout3d: int
    isave: int
    cbcvel: np.ndarray
    cbcpre: np.ndarray
    cbcsgs: np.ndarray
    bcvel: np.ndarray
    bcpre: np.ndarray
    bcsgs: np.ndarray
    bforce: np.ndarray
    is_forced: np.ndarray
    velf: np.ndarray
    dims: np.ndarray
    sgstype: str
    lwm: np.ndarray
    hwm: float
    # build-time switches of the reference (configs/flags.mk.example:105-151) are run-time here
    impdiff: int = 0          # 0 explicit, 1 _IMPDIFF, 2 _IMPDIFF + _IMPDIFF_1D

    @property
    def dl(self) -> np.ndarray:
        return self.l / self.ng.astype(np.float64)

    @property
    def dli(self) -> np.ndarray:
        return 1.0 / self.dl

    @property
    def visc(self) -> float:
        return 1.0 / self.visci

    def copy(self, **kw) -> "Case":
        d = {f.name: (getattr(self, f.name).copy() if isinstance(getattr(self, f.name), np.ndarray)
                      else getattr(self, f.name)) for f in dataclasses.fields(self)}
        d.update(kw)
        return Case(**d)


def _defaults() -> Dict[str, Any]:
    d: Dict[str, Any] = {}
    for k, (typ, shape, _) in _SCHEMA.items():
        if shape:
            d[k] = np.full(shape, " " if typ == "c" else 0,
                           dtype={"i": np.int32, "r": np.float64, "l": np.bool_, "c": "U1"}[typ], order="F")
        else:
            d[k] = {"i": 0, "r": 0.0, "l": False, "c": ""}[typ]
    d["dt_f"] = -1.0                                     # param.f90:124
    return d


def parse_text(text: str) -> Case:
    groups = _split_groups(_strip_comments(text))
    for g in ("dns", "les"):
        if g not in groups:
            raise NamelistError(f"Error reading {g} namelist: group not found")   # param.f90:135-150
    vals = _defaults()
    for g in ("dns", "les"):
        body = groups[g]
        heads = list(re.finditer(r"([A-Za-z_]\w*)\s*(\([^)]*\))?\s*=", body))
        for i, m in enumerate(heads):
            name = m.group(1).lower()
            if name not in _GROUPS[g]:
                raise NamelistError(f"Error reading {g} namelist: unknown variable {name!r}")
            typ, shape, lb = _SCHEMA[name]
            end = heads[i + 1].start() if i + 1 < len(heads) else len(body)
            toks = _tokens(body[m.end():end])
            idx = _section(m.group(2), shape, lb)
            if len(toks) > len(idx):
                raise NamelistError(f"too many values for {name}{m.group(2) or ''}")
            for t, ix in zip(toks, idx):
                v = _convert(t, typ)
                if shape:
                    vals[name][ix] = v[:1] if typ == "c" else v
                else:
                    vals[name] = v.strip() if typ == "c" else v
    return Case(**vals)


def read_input(path: str = "input.nml") -> Case:
    try:
        with open(path, "r") as fh:
            text = fh.read()
    except OSError as e:                                  # param.f90:126-132
        raise NamelistError(f"Error reading the input file: {e}") from e
    return parse_text(text)
