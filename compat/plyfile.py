"""Stand-in for the `plyfile` package, limited to what MooreThreads/LiteGS calls (see compat/README.md): vertex tables of scalar
properties, binary little-endian or ascii.  Container code: litegs_amd/io/plyformat.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from litegs_amd.io import plyformat as _fmt  # noqa: E402


class PlyProperty:
    def __init__(self, name, val_dtype):
        self.name, self.val_dtype = name, val_dtype

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.val_dtype!r})"


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data

    @staticmethod
    def describe(data, name, **_unused):
        if not isinstance(data, np.ndarray) or data.dtype.names is None:
            raise TypeError("PlyElement.describe: a numpy structured array is required")
        return PlyElement(name, data)

    @property
    def properties(self):
        return tuple(PlyProperty(n, self.data.dtype[n].str.lstrip("<>=|")) for n in self.data.dtype.names)

    @property
    def count(self):
        return self.data.shape[0]

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return self.data.shape[0]


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="<", comments=()):
        self.elements, self.text, self.byte_order, self.comments = list(elements), text, byte_order, list(comments)

    @staticmethod
    def read(stream):
        tables, comments = _fmt.read(stream if isinstance(stream, (str, os.PathLike)) else stream.name)
        return PlyData([PlyElement(n, t) for n, t in tables.items()], comments=comments)

    def write(self, stream):
        path = stream if isinstance(stream, (str, os.PathLike)) else stream.name
        _fmt.write(path, [(e.name, e.data) for e in self.elements], text=self.text, big_endian=self.byte_order == ">", comments=self.comments)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name):
        return any(e.name == name for e in self.elements)
