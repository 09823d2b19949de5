#!/usr/bin/env python3
"""How often gathers re-fetch a record under a given node / element order (CPU only):
  * elements per node-workgroup multiplicity: how many 256-node workgroups touch an element's
    per-incidence records (N1/N2/N3 gathers);
  * node fetches per element-workgroup: how many 256-element workgroups touch a node record
    (E1/E2/E3 gathers).
usage: locality.py [mesh.desmesh]   (default: the 1M-tet TetGen mesh under oracle/_ref)"""
import os, sys, numpy as np, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench, dynearthsol_amd as des
host = des.Host(cfg_text=bench.BENCH_CFG.format(res="460.0", xlen="400e3"), overrides="mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n", mesh_file=sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oracle", "_ref", "test-3d-big-460.desmesh"))
nn, ne = host.nnode, host.nelem
conn = host.array("connectivity").reshape(4, ne)
xyz = host.array("coord").reshape(3, nn)
def node_mult(order):  # order: new position -> old node id
    pos = np.empty(nn, np.int64); pos[order] = np.arange(nn)
    blk = pos[conn] // 256            # [4, ne] block of each incidence
    b = np.sort(blk, axis=0)
    distinct = 1 + (b[1:] != b[:-1]).sum(axis=0)
    return distinct.mean()
def elem_mult(eorder, norder):      # distinct (elem block, node) pairs / nn : node record fetches per node
    epos = np.empty(ne, np.int64); epos[eorder] = np.arange(ne)
    eb = np.repeat((epos // 256)[None, :], 4, axis=0).ravel()
    key = eb * nn + conn.ravel()
    return len(np.unique(key)) / nn
def morton2(a, b, bits=10):
    a = a.astype(np.int64); b = b.astype(np.int64); r = np.zeros_like(a)
    for i in range(bits):
        r |= ((a >> i) & 1) << (2*i) | ((b >> i) & 1) << (2*i+1)
    return r
def slab_order(x, y, z, W, h):
    s = np.floor(x / W).astype(np.int64)
    m = morton2(np.floor(y / h).astype(np.int64), np.floor(-z / h).astype(np.int64))
    return np.lexsort((m, s))
def morton3(x,y,z,h,bits=10):
    a=np.floor(x/h).astype(np.int64); b=np.floor(y/h).astype(np.int64); c=np.floor(-z/h).astype(np.int64); r=np.zeros_like(a)
    for i in range(bits):
        r |= ((a>>i)&1)<<(3*i) | ((b>>i)&1)<<(3*i+1) | ((c>>i)&1)<<(3*i+2)
    return r
ident = np.arange(nn)
cen = xyz[:, conn].mean(axis=1)
print("current: elements per node-block multiplicity %.2f ; node fetch per elem-block %.2f" % (node_mult(ident), elem_mult(np.arange(ne), ident)))
for W in (2000., 3000., 5000.):
    no = slab_order(xyz[0], xyz[1], xyz[2], W, 460.)
    eo = slab_order(cen[0], cen[1], cen[2], W, 460.)
    print("slab W=%g: node-block mult %.2f ; elem-block node fetch %.2f" % (W, node_mult(no), elem_mult(eo, no)))
no = np.argsort(morton3(xyz[0], xyz[1], xyz[2], 460.)); eo = np.argsort(morton3(cen[0], cen[1], cen[2], 460.))
print("full morton: node-block mult %.2f ; elem-block node fetch %.2f" % (node_mult(no), elem_mult(eo, no)))


def hilbert3(x, y, z, h, bits=10):
    """Skilling's transposed-axes Hilbert index of the cells (floor(x/h), ...)"""
    X = [np.floor(x / h).astype(np.int64), np.floor(y / h).astype(np.int64), np.floor(-z / h).astype(np.int64)]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            t = np.where(m, 0, t)
            X[0] ^= t; X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    r = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            r = (r << 1) | ((X[i] >> b) & 1)
    return r


no = np.argsort(hilbert3(xyz[0], xyz[1], xyz[2], 460.), kind="stable"); eo = np.argsort(hilbert3(cen[0], cen[1], cen[2], 460.), kind="stable")
print("hilbert: node-block mult %.2f ; elem-block node fetch %.2f" % (node_mult(no), elem_mult(eo, no)))
