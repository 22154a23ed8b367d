"""Minimal PLY reader / writer for the reference's point-cloud checkpoints (scene/gaussian_model.py:231-347).

The reference goes through the `plyfile` package (PlyData([PlyElement.describe(elements, 'vertex')]).write(path)): a
`binary_little_endian 1.0` file with ONE element `vertex` whose properties are all `float` (numpy 'f4'), in the order of
GaussianModel.construct_list_of_attributes (:231-248).  plyfile is not a dependency here; this module writes the same
bytes layout and reads binary-little-endian and ascii PLY files with scalar properties (list properties, as faces
have, are not needed and are rejected)."""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}
_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def write_vertices(path, elements):
    """elements: numpy structured array (one record per vertex)."""
    names = elements.dtype.names
    lines = ["ply", "format binary_little_endian 1.0", "element vertex %d" % len(elements)]
    for n in names:
        dt = elements.dtype[n]
        lines.append("property %s %s" % (_NAMES[dt.str[1:]], n))
    lines.append("end_header")
    le = elements.astype(elements.dtype.newbyteorder("<"), copy=False)
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(le.tobytes())


def read_vertices(path):
    """Returns the `vertex` element as a numpy structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elems, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: PLY header is not terminated" % path)
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = (tok[1], int(tok[2]), [])
                elems.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    if cur[0] == "vertex":
                        raise ValueError("%s: list property on the vertex element" % path)
                    cur[2].append(None)
                else:
                    cur[2].append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elems or elems[0][0] != "vertex":
            raise ValueError("%s: the first element must be `vertex`" % path)
        name, count, props = elems[0]
        if fmt == "binary_little_endian":
            dt = np.dtype([(n, "<" + t) for n, t in props])
            data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        elif fmt == "binary_big_endian":
            dt = np.dtype([(n, ">" + t) for n, t in props])
            data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        elif fmt == "ascii":
            dt = np.dtype([(n, t) for n, t in props])
            rows = [f.readline().split() for _ in range(count)]
            data = np.array([tuple(float(v) for v in r) for r in rows], dtype=dt)
        else:
            raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
    return data


def attribute_names(n_dc, n_rest, n_scale, n_rot, n_embedding, exclude_filter=False):
    """GaussianModel.construct_list_of_attributes (:231-248)."""
    l = ["x", "y", "z", "nx", "ny", "nz"]
    l += ["f_dc_%d" % i for i in range(n_dc)]
    l += ["f_rest_%d" % i for i in range(n_rest)]
    l.append("opacity")
    l += ["scale_%d" % i for i in range(n_scale)]
    l += ["rot_%d" % i for i in range(n_rot)]
    l += ["embedding_%d" % i for i in range(n_embedding)]
    l.append("tongue_class")
    if not exclude_filter:
        l.append("filter_3D")
    return l
