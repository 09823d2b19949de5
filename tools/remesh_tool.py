#!/usr/bin/env python3
"""A remesher for the remeshing round trip of include/des_run.h -- TEST / DEMONSTRATION TOOL, not part of
the product (which has no mesher: DESIGN.md section 7) and not a restatement of the reference's remesh()
(remeshing.cxx:2869-3189: boundary-preserving re-tetrahedralisation, barycentric / nearest-neighbour interpolation,
marker remap).  What it does is the same JOB with simpler means, so that the round trip can be exercised with a mesh
whose node and element counts CHANGE:

    remesh_tool.py <modelname> <frame> [--resolution R]

1. reads <modelname>.save.<frame> / .chkpt.<frame> (the reference's binary format, binaryio.cxx:18-41);
2. meshes the bounding box of the deformed model anew with the reference's own TetGen behind oracle/_ref/tetmesh
   (`make -C oracle ref`; uniform resolution R, default: the old mesh's median edge), lets the host library finish the mesh
   as create_new_mesh does (boundary flags, segments, renumbering along x), and drapes it over the old top and bottom
   surfaces (the depth fraction of every new node is kept between the nearest old surface heights);
3. carries the state over: nodal fields from the nearest old node, element fields from the nearest old element centroid,
   new element volumes from the new geometry, a fresh marker set (the reference's count per element) whose material is
   that of the nearest old element;
4. writes the pair as frame + 1 with its .info row (new counts), where the restart of the run expects it.

Needs numpy + scipy (cKDTree) and oracle/_ref/tetmesh; 3-D models."""
import os
import subprocess
import sys
import tempfile

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HEADER = 4096
TETMESH = os.path.join(ROOT, "oracle", "_ref", "tetmesh")


def read(fname):
    with open(fname, "rb") as f:
        raw = f.read()
    lines = raw[:HEADER].split(b"\0")[0].decode().splitlines()
    pos = [(l.split("\t")[0], int(l.split("\t")[1])) for l in lines[1:]]
    arrays = []
    for i, (name, off) in enumerate(pos):
        end = pos[i + 1][1] if i + 1 < len(pos) else len(raw)
        arrays.append((name, raw[off:end]))
    return lines[0], arrays


def write(fname, first, arrays):
    head = first + "\n"
    off = HEADER
    for name, data in arrays:
        head += "%s\t%d\n" % (name, off)
        off += len(data)
    assert len(head) < HEADER
    with open(fname, "wb") as f:
        f.write(head.encode().ljust(HEADER, b"\0"))
        for _, data in arrays:
            f.write(data)


def f64(b, *shape):
    return np.frombuffer(b, dtype=np.float64).reshape(*shape).copy()


def i32(b, *shape):
    return np.frombuffer(b, dtype=np.int32).reshape(*shape).copy()


def new_mesh(lo, hi, resolution):
    """the box [lo, hi] meshed by the reference's TetGen, finished by the host library: coord [nn,3], conn [ne,4],
    segment [nseg,3], segflag [nseg], bcflag [nn], number of top facets"""
    import dynearthsol_amd as des
    L = hi - lo
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "new.desmesh")
        g = lambda v: "%.17g" % float(v)
        subprocess.check_call([TETMESH, "--uniform", g(L[0]), g(L[1]), g(L[2]), g(resolution), raw], stderr=subprocess.DEVNULL)
        cfg = ("[sim]\nmodelname = remesh_tool\nmax_steps = 1\noutput_step_interval = 1\nis_outputting_averaged_fields = no\n[mesh]\nmeshing_option = 1\nmeshing_elem_shape = 0\n"
               "xlength = %s\nylength = %s\nzlength = %s\nresolution = %s\n[mat]\nrheology_type = elastic\nrho0 = [3000]\n" % (g(L[0]), g(L[1]), g(L[2]), g(resolution)))
        host = des.Host(cfg_text=cfg, mesh_file=raw)
    nn, ne = host.nnode, host.nelem
    coord = host.array("coord").reshape(3, nn).T.copy()
    conn = host.array("connectivity").reshape(4, ne).T.copy()
    seg = host.array("segment")
    nseg = seg.size // 3
    seg = seg.reshape(3, nseg).T.copy()
    segflag = host.array("segflag")
    bcflag = np.ctypeslib.as_array(host.mesh.bcflag, shape=(nn,)).copy()
    etop = int(host.mesh.etop)
    host.close()
    coord[:, 2] += L[2]                    # the mesher's box is [0, Lx] x [0, Ly] x [-Lz, 0]
    coord += lo
    return coord, conn.astype(np.int32), seg.astype(np.int32), segflag.astype(np.int32), bcflag.astype(np.uint32), etop


def tet_volumes(coord, conn):
    a, b, c, d = (coord[conn[:, k]] for k in range(4))
    return np.abs(np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a)) / 6.0


def main():
    argv = sys.argv[1:]
    resolution = None
    if "--resolution" in argv:
        i = argv.index("--resolution")
        resolution = float(argv[i + 1])
        del argv[i:i + 2]
    model, frame = argv[0], int(argv[1])
    if not os.access(TETMESH, os.X_OK):
        sys.exit("remesh_tool: %s is missing (make -C oracle ref, in the build container)" % TETMESH)
    first_s, save = read("%s.save.%06d" % (model, frame))
    first_c, chk = read("%s.chkpt.%06d" % (model, frame))
    if "ndims=3" not in first_s:
        sys.exit("remesh_tool: 3-D models only")
    S, Cc = dict(save), dict(chk)
    nn_old, ne_old = int(i32(S["nnode"], 1)[0]), int(i32(S["nelem"], 1)[0])
    xo = f64(S["coordinate"], nn_old, 3)
    co = i32(S["connectivity"], ne_old, 4)
    flag_o = np.frombuffer(S["bcflag"], dtype=np.uint32)
    if resolution is None:
        e = np.linalg.norm(xo[co[:, 0]] - xo[co[:, 1]], axis=1)
        resolution = float(np.median(e))
    lo, hi = xo.min(axis=0), xo.max(axis=0)
    xn, cn, seg, segflag, flag_n, etop = new_mesh(lo, hi, resolution)
    nn, ne = len(xn), len(cn)

    # drape the new box over the old top / bottom surfaces: keep every new node's depth fraction
    top_o, bot_o = xo[(flag_o & 32) != 0], xo[(flag_o & 16) != 0]
    zt = top_o[cKDTree(top_o[:, :2]).query(xn[:, :2])[1], 2]
    zb = bot_o[cKDTree(bot_o[:, :2]).query(xn[:, :2])[1], 2]
    s = (hi[2] - xn[:, 2]) / (hi[2] - lo[2])
    xn[:, 2] = zt - s * (zt - zb)

    node_of = cKDTree(xo).query(xn)[1]                               # nearest old node of every new node
    cen_o, cen_n = xo[co].mean(axis=1), xn[cn].mean(axis=1)
    elem_of = cKDTree(cen_o).query(cen_n)[1]                         # nearest old element of every new element

    # markers: the old material of an element = the majority of its markers
    nm_old = int(i32(Cc["markerset size"], 3)[0])
    mel, mmat = i32(S["markerset.elem"], nm_old), i32(S["markerset.mattype"], nm_old)
    nmat = int(mmat.max()) + 1
    counts = np.zeros((ne_old, nmat), dtype=np.int64)
    np.add.at(counts, (mel, mmat), 1)
    mat_o = counts.argmax(axis=1)
    mpe = max(1, int(round(nm_old / ne_old)))
    nm = ne * mpe
    rng = np.random.default_rng(12345)
    eta = rng.dirichlet(np.ones(4), size=nm)
    m_elem = np.repeat(np.arange(ne, dtype=np.int32), mpe)
    m_mat = mat_o[elem_of][m_elem].astype(np.int32)
    m_z = np.einsum("ij,ij->i", eta, xn[cn[m_elem], 2])

    def remap(name, data):
        """an array of the old frame on the new mesh, by its name or its size"""
        n = len(data)
        fixed = {
            "coordinate": xn, "connectivity": cn, "bcflag": flag_n, "segment": seg, "segflag": segflag,
            "nnode": np.array([nn], np.int32), "nelem": np.array([ne], np.int32), "nseg": np.array([len(seg)], np.int32),
            "volume_old": tet_volumes(xn, cn), "dv surface acc": np.zeros(etop), "markerset.eta": eta,
            "markerset.elem": m_elem, "markerset.mattype": m_mat, "markerset.id": np.arange(nm, dtype=np.int32),
            "markerset.time": np.zeros(nm), "markerset.z": m_z, "markerset.distance": np.zeros(nm), "markerset.slope": np.zeros(nm),
            "markerset.genesis": np.zeros(nm, np.int32),
            "markerset.coord": np.einsum("ij,ijk->ik", eta, xn[cn[m_elem]]),
        }
        if name in fixed:
            return np.ascontiguousarray(fixed[name]).tobytes()
        if name == "markerset size":
            return (np.array([nm], np.int32) if n == 4 else np.array([nm, nm, nm], np.int32)).tobytes()
        if name == "elemmarkers":
            em = np.zeros((ne, n // (4 * ne_old)), np.int32)
            em[np.arange(ne), mat_o[elem_of]] = mpe
            return em.tobytes()
        if name == "coord0":                                   # the new nodes' reference position: where they are, less the old offset
            off = f64(data, nn_old, 3) - xo
            return np.ascontiguousarray(xn + off[node_of]).tobytes()
        for count, idx in ((nn_old, node_of), (ne_old, elem_of)):
            for item in (8, 4):
                if n % (count * item) == 0 and n >= count * item:
                    k = n // (count * item)
                    a = np.frombuffer(data, dtype=np.float64 if item == 8 else np.int32).reshape(count, k)
                    return np.ascontiguousarray(a[idx]).tobytes()
        return data                                            # scalars: time, dt, steps, ...

    write("%s.save.%06d" % (model, frame + 1), first_s, [(n, remap(n, d)) for n, d in save])
    write("%s.chkpt.%06d" % (model, frame + 1), first_c, [(n, remap(n, d)) for n, d in chk])
    with open(model + ".info") as f:
        rows = [l for l in f if l.strip()]
    last = [r for r in rows if int(r.split()[0]) == frame][-1].rstrip("\n").split("\t")
    last[0] = "%6d" % (frame + 1)
    last[5], last[6], last[7] = "%8d" % nn, "%8d" % ne, "%8d" % len(seg)
    with open(model + ".info", "a") as f:
        f.write("\t".join(last) + "\n")
    print("remesh_tool: %d nodes / %d tets -> %d / %d (TetGen at %.6g m), %d markers, frame %d" % (nn_old, ne_old, nn, ne, resolution, nm, frame + 1))


if __name__ == "__main__":
    main()
