"""The synthetic BVH-traversal stress scene of BASELINE config 4.

A jittered, displaced n_side x n_side-quad height field (2 n_side^2 triangles; n_side = 2236 gives 10.0 M) inside
cbox-like walls under four quad lights, every surface diffuse 0.5.  Vertices come from the pcg32 stream
seed(1234, 1) (ext/pcg32/pcg32.h), generated here with numpy, so the scene is a pure function of n_side.

  c4_meshes(n_side)            the meshes as arrays
  make_c4(n_side, w, h, spp)   the scene through the host plugin classes (SceneBuilder, no files)
  write_c4(dir, n_side, ...)   the same scene as Nori files: <dir>/c4.xml + OBJ meshes (the reference's own ingest path;
                               used at reduced n_side by the parity tests and by bench.py's cpu_baseline leg)
"""
import os

import numpy as np

PCG32_MULT = 0x5851F42D4C957F2D
_M64 = (1 << 64) - 1


def pcg32_floats(initstate, initseq, n):
    """n floats of pcg32.seed(initstate, initseq).nextFloat() (ext/pcg32/pcg32.h:51-66, 101-110), vectorised: the state
    after i steps is an affine map of the seeded state, and the maps of all i are built by doubling."""
    inc = ((initseq << 1) | 1) & _M64
    s = 0
    s = (s * PCG32_MULT + inc) & _M64
    s = (s + initstate) & _M64
    s = (s * PCG32_MULT + inc) & _M64
    with np.errstate(over="ignore"):
        A = np.ones(1, np.uint64)
        Cc = np.zeros(1, np.uint64)
        a_l, c_l = PCG32_MULT, inc                       # the map of L steps, L = len(A)
        while len(A) < n:
            A = np.concatenate([A, A * np.uint64(a_l)])
            Cc = np.concatenate([Cc, Cc * np.uint64(a_l) + np.uint64(c_l)])
            a_l, c_l = (a_l * a_l) & _M64, (a_l * c_l + c_l) & _M64
        old = A[:n] * np.uint64(s) + Cc[:n]              # state BEFORE draw i (the output function reads the old state)
        xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        u = (xs >> rot) | (xs << ((np.uint32(0) - rot) & np.uint32(31)))
    return ((u >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def _quad(p):
    return np.array(p, np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32)


def c4_meshes(n_side):
    """[(name, V, F, radiance or None)]: the height field first, then back/front/left/right/ceiling walls, then four lights"""
    g = n_side + 1
    f = pcg32_floats(1234, 1, 3 * g * g).reshape(g, g, 3)
    step = 2.0 / n_side
    xs, zs = np.meshgrid(np.linspace(-1, 1, g, dtype=np.float32), np.linspace(-1, 1, g, dtype=np.float32))
    jit = (f[:, :, 0:2] - 0.5) * 0.6 * step
    jit[0, :, :] = 0; jit[-1, :, :] = 0; jit[:, 0, :] = 0; jit[:, -1, :] = 0
    V = np.stack([xs + jit[:, :, 0], 0.2 * f[:, :, 2], zs + jit[:, :, 1]], 2).reshape(-1, 3).astype(np.float32)
    i = np.arange(g * g, dtype=np.uint32).reshape(g, g)
    a, b, c, d = i[:-1, :-1].ravel(), i[:-1, 1:].ravel(), i[1:, 1:].ravel(), i[1:, :-1].ravel()
    F = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)]).astype(np.uint32)
    out = [("heightfield", V, F, None)]
    walls = {
        "back": [[-1, 0, -1], [1, 0, -1], [1, 2, -1], [-1, 2, -1]],
        "front": [[1, 0, 1], [-1, 0, 1], [-1, 2, 1], [1, 2, 1]],
        "left": [[-1, 0, 1], [-1, 0, -1], [-1, 2, -1], [-1, 2, 1]],
        "right": [[1, 0, -1], [1, 0, 1], [1, 2, 1], [1, 2, -1]],
        "ceil": [[-1, 2, -1], [1, 2, -1], [1, 2, 1], [-1, 2, 1]],
    }
    for k, p in walls.items():
        v, fi = _quad(p)
        out.append((k, v, fi, None))
    for li, (cx, cz) in enumerate([(-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5)]):
        s = 0.15
        # wound so that the geometric normal (p1 - p0) x (p2 - p0) points DOWN: area emitters are one-sided (src/arealight.cpp:41-54).
        # (Until round 2 the quads were wound the other way round and lit only the ceiling 2 cm above them: every emitter sample
        # taken from below evaluated to zero and the scene was lit by the ceiling's reflection alone.)
        v, fi = _quad([[cx - s, 1.98, cz - s], [cx + s, 1.98, cz - s], [cx + s, 1.98, cz + s], [cx - s, 1.98, cz + s]])
        out.append(("light%d" % li, v, fi, (15.0, 15.0, 15.0)))
    return out


C4_CAMERA = dict(origin=(0.0, 1.0, 0.95), target=(0.0, 0.1, -0.3), up=(0.0, 1.0, 0.0), fov=60.0)


def make_c4(n_side, width, height, spp):
    """the scene through the host plugin classes; returns a HostScene (the tree is built by Scene::activate unless
    NORI_HIP_BVH=device)"""
    import nori_hip as NH
    sb = NH.SceneBuilder()
    for name, V, F, rad in c4_meshes(n_side):
        sb.add_mesh(name, V, F, color=(0.5, 0.5, 0.5), radiance=rad)
    sb.set_camera(C4_CAMERA["origin"], C4_CAMERA["target"], C4_CAMERA["up"], C4_CAMERA["fov"], width, height)
    return sb.finish(spp=spp)


def write_c4(directory, n_side, width, height, spp):
    """c4.xml + one OBJ per mesh in `directory`; returns the XML path.  Floats are written with 9 significant digits, which
    round-trips fp32, so every loader of the files sees the vertices of c4_meshes() bit for bit."""
    os.makedirs(directory, exist_ok=True)
    xml = ['<?xml version="1.0" encoding="utf-8"?>', "<scene>", '\t<integrator type="path_mis"/>',
           '\t<camera type="perspective">', '\t\t<float name="fov" value="%.9g"/>' % C4_CAMERA["fov"], '\t\t<transform name="toWorld">',
           '\t\t\t<lookat target="%s" origin="%s" up="%s"/>' % tuple(", ".join("%.9g" % x for x in C4_CAMERA[k]) for k in ("target", "origin", "up")),
           "\t\t</transform>", '\t\t<integer name="width" value="%d"/>' % width, '\t\t<integer name="height" value="%d"/>' % height, "\t</camera>",
           '\t<sampler type="independent">', '\t\t<integer name="sampleCount" value="%d"/>' % spp, "\t</sampler>"]
    for name, V, F, rad in c4_meshes(n_side):
        with open(os.path.join(directory, name + ".obj"), "w") as fh:
            np.savetxt(fh, V, fmt="v %.9g %.9g %.9g")
            np.savetxt(fh, F.astype(np.int64) + 1, fmt="f %d %d %d")
        xml += ['\t<mesh type="obj">', '\t\t<string name="filename" value="%s.obj"/>' % name,
                '\t\t<bsdf type="diffuse">', '\t\t\t<color name="albedo" value="0.5 0.5 0.5"/>', "\t\t</bsdf>"]
        if rad is not None:
            xml += ['\t\t<emitter type="area">', '\t\t\t<color name="radiance" value="%g %g %g"/>' % rad, "\t\t</emitter>"]
        xml += ["\t</mesh>"]
    xml += ["</scene>", ""]
    path = os.path.join(directory, "c4.xml")
    with open(path, "w") as fh:
        fh.write("\n".join(xml))
    return path
