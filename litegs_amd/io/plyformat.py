"""Minimal PLY container (header + fixed-size scalar properties), numpy only.

The reference reads and writes point clouds through the third-party ``plyfile`` package (litegs/io_manager/ply.py:4,
litegs/io_manager/colmap.py:6), which is not in this image.  This module is the container format itself, written from the
PLY specification: ``ply / format {ascii|binary_little_endian|binary_big_endian} 1.0 / element <name> <count> /
property <type> <name> ... / end_header`` followed by the element tables.  List properties (faces) are not needed on this path
and are rejected.  An element is a numpy structured array -- one ``tofile``/``fromfile`` per element, no per-vertex Python.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

# PLY scalar type names (both spellings of the specification) -> numpy type codes without byte order
_PLY_TO_NP = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def header_bytes(elements: List[Tuple[str, np.ndarray]], fmt: str = "binary_little_endian", comments: List[str] = ()) -> bytes:
    lines = ["ply", f"format {fmt} 1.0"]
    lines += [f"comment {c}" for c in comments]
    for name, table in elements:
        lines.append(f"element {name} {table.shape[0]}")
        for field in table.dtype.names:
            code = table.dtype[field].str.lstrip("<>=|")
            if code not in _NP_TO_PLY:
                raise ValueError(f"property {field}: dtype {table.dtype[field]} has no PLY scalar type")
            lines.append(f"property {_NP_TO_PLY[code]} {field}")
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode("ascii")


def write(path: str, elements: List[Tuple[str, np.ndarray]], text: bool = False, big_endian: bool = False, comments: List[str] = ()) -> None:
    """elements: [(name, structured array)] in file order."""
    fmt = "ascii" if text else ("binary_big_endian" if big_endian else "binary_little_endian")
    with open(path, "wb") as f:
        f.write(header_bytes(elements, fmt, comments))
        for _, table in elements:
            if text:
                for row in table:
                    f.write((" ".join(repr(v.item()) if v.dtype.kind == "f" else str(v.item()) for v in row) + "\n").encode("ascii"))
            else:
                order = ">" if big_endian else "<"
                f.write(np.ascontiguousarray(table.astype(table.dtype.newbyteorder(order), copy=False)).tobytes())


def read_header(f) -> Tuple[str, List[Tuple[str, int, List[Tuple[str, str]]]], List[str]]:
    """-> (format, [(element name, count, [(property name, numpy code)])], comments); leaves f at the first body byte."""
    if f.readline().strip() != b"ply":
        raise ValueError("not a PLY file")
    fmt, elements, comments = None, [], []
    while True:
        line = f.readline()
        if not line:
            raise ValueError("PLY header without end_header")
        tok = line.decode("ascii", "replace").strip().split()
        if not tok:
            continue
        if tok[0] == "end_header":
            break
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] in ("comment", "obj_info"):
            comments.append(" ".join(tok[1:]))
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError("PLY list properties are not supported (point clouds only)")
            if tok[1] not in _PLY_TO_NP:
                raise ValueError(f"unknown PLY property type {tok[1]}")
            elements[-1][2].append((tok[2], _PLY_TO_NP[tok[1]]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"unknown PLY format {fmt}")
    return fmt, elements, comments


def read(path: str) -> Tuple[Dict[str, np.ndarray], List[str]]:
    """-> ({element name: structured array (native byte order)}, comments); element order = dict order."""
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        fmt, elements, comments = read_header(f)
        for name, count, props in elements:
            native = np.dtype([(p, c) for p, c in props])
            if fmt == "ascii":
                table = np.empty((count,), dtype=native)
                for i in range(count):
                    tok = f.readline().split()
                    if len(tok) < len(props):
                        raise ValueError(f"PLY element {name}: row {i} is short")
                    table[i] = tuple(np.dtype(c).type(t.decode()) if np.dtype(c).kind == "f" else int(t) for (p, c), t in zip(props, tok))
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                disk = np.dtype([(p, order + c) for p, c in props])
                raw = np.fromfile(f, dtype=disk, count=count)
                if raw.shape[0] != count:
                    raise ValueError(f"PLY element {name}: file is truncated ({raw.shape[0]} of {count} rows)")
                table = raw.astype(native, copy=False)
            out[name] = table
    return out, comments
