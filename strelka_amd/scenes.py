"""Procedural stand-ins for the BASELINE.json scenes (none of the USD assets exist in the reference tree or in the
container; SURVEY.md section 8d fixes these recipes).  All generators are seeded (numpy RandomState = MT19937) and emit
the flat ``oka::Scene`` arrays through strelka_amd.scene.Scene, exactly what the renderer would receive from HdStrelka.

    cornell_box()          C2: classic Cornell box, Lambert only, 1 rect light
    kitchen_standin()      C3/C4: >= 1.0 M unique triangles, >= 2000 instances of >= 150 meshes, mixed materials
    kitchen_architectural() C3, less forgiving: big flat quads in two triangles, long thin triangles, nested cabinets, no mesh sharing
    hair_standin()         C5: 100 k strands x 16 control points (+2 phantom), hair BSDF
    coffeemaker_standin()  C1: ~50 k-triangle lathe object on a ground plane
"""
import math

import numpy as np

from . import scene as S


def _grid_mesh(fn, nu, nv, wrap_u=False):
    """Triangulated parametric surface p = fn(u, v), u,v in [0,1]; returns (positions, triangles)."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    pos = fn(uu.reshape(-1), vv.reshape(-1)).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i, j = i.reshape(-1), j.reshape(-1)
    a = i * (nv + 1) + j
    b = (i + 1) * (nv + 1) + j
    tris = np.stack([np.stack([a, b, a + 1], 1), np.stack([a + 1, b, b + 1], 1)], 1).reshape(-1, 3)
    return pos, tris


def _sphere_fn(rs, bump):
    ph = rs.uniform(0, 2 * math.pi, 6)
    fr = rs.randint(2, 7, 6)

    def fn(u, v):
        theta = v * math.pi
        phi = u * 2 * math.pi
        r = 1.0 + bump * (np.sin(fr[0] * phi + ph[0]) * np.sin(fr[1] * theta + ph[1]) +
                          0.5 * np.sin(fr[2] * phi + ph[2]) * np.sin(fr[3] * theta + ph[3]))
        st = np.sin(theta)
        return np.stack([r * st * np.cos(phi), r * np.cos(theta), r * st * np.sin(phi)], 1)

    return fn


def _torus_fn(rs):
    r1 = rs.uniform(0.2, 0.45)

    def fn(u, v):
        a, b = u * 2 * math.pi, v * 2 * math.pi
        return np.stack([(1 + r1 * np.cos(b)) * np.cos(a), r1 * np.sin(b), (1 + r1 * np.cos(b)) * np.sin(a)], 1)

    return fn


def _lathe_fn(rs):
    k = rs.uniform(0.3, 1.0, 4)
    ph = rs.uniform(0, 2 * math.pi, 4)

    def fn(u, v):
        a = u * 2 * math.pi
        prof = 0.55 + 0.25 * np.sin(k[0] * 6 * v + ph[0]) + 0.12 * np.sin(k[1] * 14 * v + ph[1])
        prof = prof * np.sin(np.clip(v, 0.0, 1.0) * math.pi) ** 0.35  # closed at both ends
        return np.stack([prof * np.cos(a), 2.0 * v - 1.0, prof * np.sin(a)], 1)

    return fn


def _box_mesh(lo, hi, inward=False, skip_bottom=False):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
    q = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]  # outward CCW
    tris = []
    for fi, (a, b, cc, d) in enumerate(q):
        if skip_bottom and fi == 2:
            continue
        tris += [(a, b, cc), (a, cc, d)] if not inward else [(a, cc, b), (a, d, cc)]
    return c, np.array(tris)


def _quad(p0, p1, p2, p3):
    return np.array([p0, p1, p2, p3], np.float32), np.array([(0, 1, 2), (0, 2, 3)])


def _add_mesh(sc, pos, tris):
    vb, ib = S.deindex(pos, tris)
    return sc.createMesh(vb, ib)


# ----------------------------------------------------------------------------------------------------------
def cornell_box():
    """C2: 5 walls + short + tall block = 30 triangles in 3 meshes (white, red, green) + 1 rect-light
    proxy = 4 instances; light 0.26 x 0.21 at the ceiling, colour (17, 12, 4); camera fov 39.3 deg; Lambert only."""
    sc = S.Scene()
    white = sc.addMaterial(S.MAT_DIFFUSE, (0.725, 0.71, 0.68))
    red = sc.addMaterial(S.MAT_DIFFUSE, (0.63, 0.065, 0.05))
    green = sc.addMaterial(S.MAT_DIFFUSE, (0.14, 0.45, 0.091))
    # unit box [-1,1]^3 open towards +z; normals face inward
    P, T = [], []

    def add(pos, tris):
        off = sum(len(p) for p in P)
        P.append(pos)
        T.append(tris + off)

    add(*_quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)))  # floor, normal +y
    add(*_quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)))  # ceiling, normal -y
    add(*_quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)))  # back wall, normal +z

    def block(cx, cz, w, h, ang):  # 5 faces (no bottom), outward normals
        pos, tris = _box_mesh((-w, 0, -w), (w, h, w), skip_bottom=True)
        r = S.rotate((0, 1, 0), ang)[:3, :3]
        pos = (pos.astype(np.float64) @ r.T + np.array([cx, -1.0, cz])).astype(np.float32)
        return pos, tris

    add(*block(0.33, 0.35, 0.3, 0.6, math.radians(-17)))  # short block
    add(*block(-0.33, -0.3, 0.3, 1.2, math.radians(17)))  # tall block
    pw, tw = np.concatenate(P), np.concatenate(T)
    m_white = _add_mesh(sc, pw, tw)  # 6 + 10 + 10 = 26 triangles
    m_red = _add_mesh(sc, *_quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)))  # left wall, normal +x
    m_green = _add_mesh(sc, *_quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)))  # right wall, normal -x
    I = np.eye(4)
    sc.createInstance(S.INSTANCE_MESH, m_white, white, I)
    sc.createInstance(S.INSTANCE_MESH, m_red, red, I)
    sc.createInstance(S.INSTANCE_MESH, m_green, green, I)
    # rect light just under the ceiling, emitting downwards: local -Z is the emitting side (Lights.h:54-62),
    # so rotate local +Z to world +Y
    xf = S.translate((0.0, 0.995, 0.0)) @ S.rotate((1, 0, 0), math.radians(-90))
    sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 0.52, "height": 0.42, "color": (17.0, 12.0, 4.0),
                    "intensity": 1.0})
    cam = S.Camera(fov=39.3)
    cam.lookAt((0.0, 0.0, 3.9), (0.0, 0.0, 0.0))
    sc.addCamera(cam)
    return sc


def kitchen_standin(seed=1234, n_meshes=150, n_instances=2000, tri_lo=200, tri_hi=50000, target_tris=None):
    """C3 "kitchen stand-in" (SURVEY 8d): room 10 x 6 x 4 units, >= 1.0 M unique triangles in >= 2000 instances of
    >= 150 meshes (sizes log-uniform tri_lo..tri_hi, de-indexed like Mesh.cpp:140-178), 4 rect lights + 1 distant
    (half-angle 5 deg), materials 60 % diffuse / 25 % glossy (roughness U[0.05,0.6]) / 10 % metal / 5 % glass."""
    rs = np.random.RandomState(seed)
    sc = S.Scene()
    # materials: 0 = default white
    sc.addMaterial(S.MAT_DIFFUSE, (0.8, 0.8, 0.8))
    n_mat = 64
    kinds = rs.choice(4, size=n_mat, p=[0.60, 0.25, 0.10, 0.05])
    mats = []
    for k in kinds:
        col = tuple(rs.uniform(0.15, 0.9, 3))
        if k == 0:
            mats.append(sc.addMaterial(S.MAT_DIFFUSE, col))
        elif k == 1:
            mats.append(sc.addMaterial(S.MAT_PBR, col, roughness=rs.uniform(0.05, 0.6), metallic=0.0, specular=0.5))
        elif k == 2:
            mats.append(sc.addMaterial(S.MAT_PBR, col, roughness=rs.uniform(0.1, 0.5), metallic=1.0, specular=0.5))
        else:
            mats.append(sc.addMaterial(S.MAT_GLASS, (0.95, 0.97, 0.98), roughness=0.0, ior=1.5))  # clear glass (frosting_roughness 0)
    glass_mats = [m for m, k in zip(mats, kinds) if k == 3]
    # room shell (inward-facing), 12 triangles
    RX, RY, RZ = 5.0, 2.0, 3.0  # half extents: room 10 x 4 (height) x 6
    room = _add_mesh(sc, *_box_mesh((-RX, 0.0, -RZ), (RX, 2 * RY, RZ), inward=True))
    sc.createInstance(S.INSTANCE_MESH, room, 0, np.eye(4))
    # meshes
    sizes = np.exp(rs.uniform(math.log(tri_lo), math.log(tri_hi), n_meshes))
    if target_tris:
        sizes *= target_tris / sizes.sum()
    mesh_ids, closed = [], []
    for i, nt in enumerate(sizes):
        kind = i % 3
        nt = max(int(nt), 16)
        nu = max(4, int(round(math.sqrt(nt / 2.0 * 1.5))))
        nv = max(3, int(round(nt / 2.0 / nu)))
        if kind == 0:
            pos, tris = _grid_mesh(_sphere_fn(rs, rs.uniform(0.03, 0.2)), nu, nv)
        elif kind == 1:
            pos, tris = _grid_mesh(_torus_fn(rs), nu, nv)
        else:
            pos, tris = _grid_mesh(_lathe_fn(rs), nu, nv)
        # orient consistently outward (parametric grids above come out inward for spheres): flip by signed volume
        p = pos.astype(np.float64)
        vol = np.einsum("ij,ij->i", p[tris[:, 0]], np.cross(p[tris[:, 1]], p[tris[:, 2]])).sum()
        if vol < 0:
            tris = tris[:, ::-1]
        mesh_ids.append(_add_mesh(sc, pos, tris))
        closed.append(kind != 1 or True)
    # instances: one per cell of a jittered layout -- a floor grid plus four shelf levels (two rows deep) along the
    # walls -- so that objects touch but rarely interpenetrate (keeps the instance boxes of the TLAS from piling up)
    cell = 0.2
    cells = []
    xs = np.arange(-RX + 0.3, RX - 0.3 + 1e-6, cell)
    zs = np.arange(-RZ + 0.3, RZ - 0.3 + 1e-6, cell)
    for x in xs:
        for z in zs:
            cells.append((x, 0.0, z))
    for lvl in (0.9, 1.6, 2.3, 3.0):
        for row in (0.12, 0.32):
            for x in xs:
                cells.append((x, lvl, -RZ + row))
                cells.append((x, lvl, RZ - row))
            for z in zs:
                cells.append((-RX + row, lvl, z))
                cells.append((RX - row, lvl, z))
    cells = np.array(cells)
    order = rs.permutation(len(cells))
    assert len(cells) >= n_instances, (len(cells), n_instances)
    for k in range(n_instances):
        m = mesh_ids[k % n_meshes] if k < n_meshes else mesh_ids[rs.randint(n_meshes)]
        sx = rs.uniform(0.05, 0.1) * (1.0 if rs.rand() < 0.97 else 3.0)
        sy = sx * rs.uniform(0.8, 1.5)
        cx, cy, cz = cells[order[k]]
        x = cx + rs.uniform(-0.03, 0.03)
        z = cz + rs.uniform(-0.03, 0.03)
        y = cy + 1.02 * sy  # meshes span about [-1, 1] in y: rest them on the floor / shelf
        ang = rs.uniform(0, 2 * math.pi)
        xf = S.translate((x, y, z)) @ S.rotate((0, 1, 0), ang) @ S.scale((sx, sy, sx))
        mat = mats[rs.randint(n_mat)]
        sc.createInstance(S.INSTANCE_MESH, m, mat, xf)
    # shelf boards (thin boxes) so that the shelf objects cast and receive contact shadows
    board_pos, board_tris = _box_mesh((-1, -1, -1), (1, 1, 1))
    board = _add_mesh(sc, board_pos, board_tris)
    for lvl in (0.9, 1.6, 2.3, 3.0):
        for (bx, bz, hx, hz) in [(0.0, -RZ + 0.22, RX - 0.05, 0.22), (0.0, RZ - 0.22, RX - 0.05, 0.22),
                                 (-RX + 0.22, 0.0, 0.22, RZ - 0.05), (RX - 0.22, 0.0, 0.22, RZ - 0.05)]:
            sc.createInstance(S.INSTANCE_MESH, board, 0, S.translate((bx, lvl - 0.012, bz)) @ S.scale((hx, 0.01, hz)))
    # lights: 4 ceiling rect lights facing down + 1 distant light through the (imaginary) window
    for lx, lz in [(-2.5, -1.2), (2.5, -1.2), (-2.5, 1.2), (2.5, 1.2)]:
        xf = S.translate((lx, 2 * RY - 0.02, lz)) @ S.rotate((1, 0, 0), math.radians(-90))
        sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 1.2, "height": 0.8,
                        "color": (1.0, 0.96, 0.9), "intensity": 30.0})
    # distant light (half-angle 5 deg): direction of travel = xform * (0,0,-1)
    xf = S.rotate((0, 1, 0), math.radians(30)) @ S.rotate((1, 0, 0), math.radians(-55))
    sc.createLight({"type": 3, "xform": xf, "useXform": True, "halfAngle": math.radians(5.0), "color": (1.0, 0.95, 0.85),
                    "intensity": 2.0, "radius": 0.0})
    cam = S.Camera(fov=60.0)
    cam.lookAt((-4.2, 2.3, 2.3), (1.2, 0.7, -1.2))
    sc.addCamera(cam)
    return sc


def kitchen_architectural(seed=4321, n_objects=1500, target_tris=1.6e6):
    """A less forgiving stand-in for C3: what a USD kitchen looks like after HdStrelka's bake (Mesh.cpp:123-179:
    triangulated, smooth normals, one mesh per instance) rather than a room full of round blobs --
      * large flat quads cut into TWO triangles each: walls, floor, ceiling, counter tops, cabinet panels, doors, shelf boards (triangles
        metres long next to objects centimetres across);
      * long thin triangles: legs, rails, pipes, handles and blind slats -- cylinders with many segments around and ONE along (aspect
        ratios of 50 : 1 and more), thin boxes;
      * nested furniture-scale containment: cabinets (open boxes of thin panels) hold shelf boards, the shelves hold crockery; a table
        stands over chairs; every object's box sits inside its shelf's, inside its cabinet's, inside the room's;
      * no mesh sharing: every instance has its own mesh, so everything is baked to world space.
    ~1.6 M triangles, ~2 100 instances, the same lights, camera and material mix as kitchen_standin()."""
    rs = np.random.RandomState(seed)
    sc = S.Scene()
    sc.addMaterial(S.MAT_DIFFUSE, (0.8, 0.8, 0.8))
    n_mat = 64
    kinds = rs.choice(4, size=n_mat, p=[0.60, 0.25, 0.10, 0.05])
    mats = []
    for k in kinds:
        col = tuple(rs.uniform(0.15, 0.9, 3))
        if k == 0:
            mats.append(sc.addMaterial(S.MAT_DIFFUSE, col))
        elif k == 1:
            mats.append(sc.addMaterial(S.MAT_PBR, col, roughness=rs.uniform(0.05, 0.6), metallic=0.0, specular=0.5))
        elif k == 2:
            mats.append(sc.addMaterial(S.MAT_PBR, col, roughness=rs.uniform(0.1, 0.5), metallic=1.0, specular=0.5))
        else:
            mats.append(sc.addMaterial(S.MAT_GLASS, (0.95, 0.97, 0.98), roughness=0.0, ior=1.5))
    wood = [m for m, k in zip(mats, kinds) if k in (0, 1)]
    metal = [m for m, k in zip(mats, kinds) if k == 2] or wood
    RX, RY, RZ = 5.0, 2.0, 3.0
    unit_box = _box_mesh((-1, -1, -1), (1, 1, 1))

    def box(lo, hi, mat):
        """one instance of its own 12-triangle mesh (a unit box scaled into place: the mesh is NOT shared)"""
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        m = _add_mesh(sc, *unit_box)
        sc.createInstance(S.INSTANCE_MESH, m, mat, S.translate(tuple((lo + hi) / 2)) @ S.scale(tuple((hi - lo) / 2)))

    def rod(p0, p1, radius, mat, around=24):
        """a cylinder with `around` segments around and ONE along its axis (2 x around long thin triangles + two fans of caps)"""
        p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
        ax = p1 - p0
        ln = np.linalg.norm(ax)
        ax /= ln
        ref = np.array([0.0, 1.0, 0.0]) if abs(ax[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
        e1 = np.cross(ax, ref)
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(ax, e1)
        ang = np.arange(around) * (2 * math.pi / around)
        ring = np.cos(ang)[:, None] * e1 + np.sin(ang)[:, None] * e2
        pos = np.concatenate([p0 + radius * ring, p1 + radius * ring, [p0], [p1]]).astype(np.float32)
        tris = []
        for k in range(around):
            k1 = (k + 1) % around
            tris += [(k, k1, around + k), (k1, around + k1, around + k), (2 * around, k1, k), (2 * around + 1, around + k, around + k1)]
        sc.createInstance(S.INSTANCE_MESH, _add_mesh(sc, pos, np.array(tris)), mat, np.eye(4))

    # room shell: six quads = 12 triangles, metres long
    room = _add_mesh(sc, *_box_mesh((-RX, 0.0, -RZ), (RX, 2 * RY, RZ), inward=True))
    sc.createInstance(S.INSTANCE_MESH, room, 0, np.eye(4))
    # cabinets along the two long walls (floor units with a counter top, wall units above), and along one short wall
    slots = []  # (lo, hi) of every shelf compartment an object may stand in
    T = 0.018  # panel thickness

    def cabinet(x0, x1, y0, y1, z_wall, depth, side, shelves):
        """an open box of five thin panels facing the room, `shelves` boards inside, two door panels hanging half open, a rail handle"""
        zf = z_wall + side * depth  # front
        za, zb = min(z_wall, zf), max(z_wall, zf)
        mat = wood[rs.randint(len(wood))]
        box((x0, y0, za), (x0 + T, y1, zb), mat)
        box((x1 - T, y0, za), (x1, y1, zb), mat)
        box((x0, y0, za), (x1, y0 + T, zb), mat)
        box((x0, y1 - T, za), (x1, y1, zb), mat)
        box((x0, y0, z_wall - (T if side > 0 else 0)), (x1, y1, z_wall + (T if side < 0 else 0) + (T if side > 0 else 0)), mat)  # back panel
        ys = np.linspace(y0, y1, shelves + 2)
        for k in range(shelves + 1):
            if k > 0:
                box((x0 + T, ys[k] - T / 2, za + 0.01), (x1 - T, ys[k] + T / 2, zb - 0.01), mat)
            slots.append(((x0 + T + 0.01, ys[k] + T, za + 0.03), (x1 - T - 0.01, ys[k + 1] - T, zb - 0.03)))
        # door: a thin panel swung open about its outer edge + a rail handle (long thin triangles)
        w = (x1 - x0) / 2
        if side < 0 and x0 < -2.4:
            return  # (the units next to the camera have lost their doors: an open one would fill the picture)
        for hinge, sgn in ((x0, 1.0), (x1, -1.0)):
            ang = rs.uniform(1.1, 2.2)  # (swung wide open: the shelves' contents are in view)
            dx, dz = sgn * w * math.cos(ang), side * w * math.sin(ang)
            m = _add_mesh(sc, *unit_box)
            xf = (S.translate((hinge + dx / 2, (y0 + y1) / 2, zf + dz / 2)) @ S.rotate((0, 1, 0), -sgn * side * ang) @
                  S.scale((w / 2, (y1 - y0) / 2 - 0.005, T / 2)))
            sc.createInstance(S.INSTANCE_MESH, m, mat, xf)
            hx, hz = hinge + 0.85 * dx, zf + 0.85 * dz + side * 0.03
            rod((hx, y0 + 0.25 * (y1 - y0), hz), (hx, y0 + 0.75 * (y1 - y0), hz), 0.006, metal[rs.randint(len(metal))], around=16)

    xs = np.arange(-RX + 0.1, RX - 0.7, 0.8)
    for side, zw in ((1.0, -RZ + 0.02), (-1.0, RZ - 0.02)):
        for x0 in xs:
            cabinet(x0, x0 + 0.78, 0.1, 0.9, zw, 0.6, side, 1)  # floor unit
            cabinet(x0, x0 + 0.78, 1.5, 2.4, zw, 0.35, side, 2)  # wall unit
        # one counter top over the whole run: a slab 9 m long, 12 triangles; a splash-back behind it; a toe-kick below
        box((-RX + 0.08, 0.9, min(zw, zw + side * 0.64)), (RX - 0.7 + 0.1, 0.94, max(zw, zw + side * 0.64)), wood[rs.randint(len(wood))])
        for x0 in xs:  # things standing on the counter
            slots.append(((x0 + 0.02, 0.94, min(zw + side * 0.05, zw + side * 0.6)), (x0 + 0.76, 1.45, max(zw + side * 0.05, zw + side * 0.6))))
        box((-RX + 0.08, 0.94, min(zw, zw + side * 0.012)), (RX - 0.6, 1.5, max(zw, zw + side * 0.012)), mats[rs.randint(n_mat)])
        box((-RX + 0.1, 0.0, min(zw + side * 0.05, zw + side * 0.55)), (RX - 0.62, 0.1, max(zw + side * 0.05, zw + side * 0.55)), wood[rs.randint(len(wood))])
        # a pipe run under the ceiling and a curtain of blind slats in front of the wall units' gap: long thin geometry
        rod((-RX + 0.1, 3.7, zw + side * 0.15), (RX - 0.1, 3.7, zw + side * 0.15), 0.03, metal[rs.randint(len(metal))], around=32)
        rod((-RX + 0.1, 3.55, zw + side * 0.3), (RX - 0.1, 3.55, zw + side * 0.3), 0.015, metal[rs.randint(len(metal))], around=24)
    for k in range(60):  # blind slats across the far short wall (a window): 60 boards 5.6 m x 2.5 cm x 2 mm
        y = 1.0 + k * 0.03
        m = _add_mesh(sc, *unit_box)
        sc.createInstance(S.INSTANCE_MESH, m, mats[rs.randint(n_mat)],
                          S.translate((RX - 0.06, y, 0.0)) @ S.rotate((0, 0, 1), 0.5) @ S.scale((0.0125, 0.001, RZ - 0.2)))
    # table + chairs in the middle: a top over four legs, chairs of rods and thin boards pushed under it
    tx, tz = 0.0, 0.0
    box((tx - 1.1, 0.74, tz - 0.5), (tx + 1.1, 0.78, tz + 0.5), wood[rs.randint(len(wood))])
    for sx_ in (-1.0, 1.0):
        for sz_ in (-1.0, 1.0):
            rod((tx + sx_ * 1.02, 0.0, tz + sz_ * 0.42), (tx + sx_ * 1.02, 0.74, tz + sz_ * 0.42), 0.025, wood[rs.randint(len(wood))])
    slots.append(((tx - 1.05, 0.78, tz - 0.45), (tx + 1.05, 1.1, tz + 0.45)))
    for k in range(6):
        cx, cz = tx - 0.8 + 0.8 * (k % 3), tz + (0.75 if k < 3 else -0.75)
        mat = wood[rs.randint(len(wood))]
        box((cx - 0.2, 0.44, cz - 0.2), (cx + 0.2, 0.46, cz + 0.2), mat)
        for sx_ in (-1.0, 1.0):
            for sz_ in (-1.0, 1.0):
                top = 0.95 if (sz_ > 0) == (k < 3) else 0.44
                rod((cx + sx_ * 0.18, 0.0, cz + sz_ * 0.18), (cx + sx_ * 0.18, top, cz + sz_ * 0.18), 0.012, mat, around=12)
        zb = cz + (0.18 if k < 3 else -0.18)
        for yy in (0.6, 0.72, 0.84):
            box((cx - 0.19, yy, zb - 0.008), (cx + 0.19, yy + 0.05, zb + 0.008), mat)
    # crockery: every object its own mesh (the blobs of kitchen_standin, scaled to fit a compartment), several to a compartment
    sizes = np.exp(rs.uniform(math.log(200), math.log(5000), n_objects))
    sizes *= target_tris / sizes.sum()
    order = rs.permutation(len(slots))
    for i, nt in enumerate(sizes):
        lo, hi = slots[order[i % len(slots)]]
        lo, hi = np.asarray(lo), np.asarray(hi)
        kind = i % 3
        nt = max(int(nt), 16)
        nu = max(4, int(round(math.sqrt(nt / 2.0 * 1.5))))
        nv = max(3, int(round(nt / 2.0 / nu)))
        fn = _sphere_fn(rs, rs.uniform(0.03, 0.2)) if kind == 0 else (_torus_fn(rs) if kind == 1 else _lathe_fn(rs))
        pos, tris = _grid_mesh(fn, nu, nv)
        pp = pos.astype(np.float64)
        if np.einsum("ij,ij->i", pp[tris[:, 0]], np.cross(pp[tris[:, 1]], pp[tris[:, 2]])).sum() < 0:
            tris = tris[:, ::-1]
        h = float(hi[1] - lo[1])
        sy = min(rs.uniform(0.04, 0.11), 0.45 * h)
        sx = min(sy * rs.uniform(0.5, 1.0), 0.2 * float(min(hi[0] - lo[0], hi[2] - lo[2])))
        x = rs.uniform(lo[0] + sx * 1.5, hi[0] - sx * 1.5)
        z = rs.uniform(lo[2] + sx * 1.5, hi[2] - sx * 1.5)
        xf = S.translate((x, lo[1] + 1.02 * sy, z)) @ S.rotate((0, 1, 0), rs.uniform(0, 2 * math.pi)) @ S.scale((sx, sy, sx))
        sc.createInstance(S.INSTANCE_MESH, _add_mesh(sc, pos, tris), mats[rs.randint(n_mat)], xf)
    for lx, lz in [(-2.5, -1.2), (2.5, -1.2), (-2.5, 1.2), (2.5, 1.2)]:
        xf = S.translate((lx, 2 * RY - 0.02, lz)) @ S.rotate((1, 0, 0), math.radians(-90))
        sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 1.2, "height": 0.8,
                        "color": (1.0, 0.96, 0.9), "intensity": 30.0})
    xf = S.rotate((0, 1, 0), math.radians(30)) @ S.rotate((1, 0, 0), math.radians(-55))
    sc.createLight({"type": 3, "xform": xf, "useXform": True, "halfAngle": math.radians(5.0), "color": (1.0, 0.95, 0.85),
                    "intensity": 2.0, "radius": 0.0})
    cam = S.Camera(fov=60.0)
    cam.lookAt((-4.2, 2.3, 2.3), (1.2, 0.7, -1.2))
    sc.addCamera(cam)
    return sc


def hair_standin(seed=77, n_strands=100000, n_cp=16, n_prims=1, prim_offset=0.0, n_moved=None, shared_xform=False):
    """C5 "hair stand-in" (SURVEY 8d): strands rooted on a unit sphere, length U[0.3,0.6], gravity-bent, root radius
    4e-4 -> tip 1e-4, +2 phantom points per strand (BasisCurves.cpp:189-232); one ~5 k-triangle scalp mesh; 2 rect
    lights; hair BSDF.
    n_prims > 1: the SAME strands handed over as n_prims curve sets, one instance each (a groom authored as several HdBasisCurves
    rprims: RenderPass.cpp:281 makes one oka::Curve + one instance per rprim) -- cut by azimuth around the head, so that a prim is a
    region of the scalp; prim_offset != 0 gives prim k (all of them, or the first n_moved) the translation (k + 1) * prim_offset along x
    instead of the identity (the strands are moved back by the same amount first, in float64, so the picture stays what it was to ~1e-7);
    shared_xform: the moved prims all sit under ONE transform -- a rotation about y + the translation: the groom of a character under an Xform."""
    rs = np.random.RandomState(seed)
    sc = S.Scene()
    sc.addMaterial(S.MAT_DIFFUSE, (0.6, 0.5, 0.45))
    hair = sc.addHairMaterial((0.35, 0.2, 0.1), roughness_r=0.3, roughness_n=0.3)
    scalp_pos, scalp_tris = _grid_mesh(_sphere_fn(rs, 0.0), 64, 40)
    p = scalp_pos.astype(np.float64)
    if np.einsum("ij,ij->i", p[scalp_tris[:, 0]], np.cross(p[scalp_tris[:, 1]], p[scalp_tris[:, 2]])).sum() < 0:
        scalp_tris = scalp_tris[:, ::-1]
    scalp = _add_mesh(sc, scalp_pos * 0.995, scalp_tris)
    sc.createInstance(S.INSTANCE_MESH, scalp, 0, np.eye(4))
    # roots on the upper 70 % of the sphere
    z = rs.uniform(-0.4, 1.0, n_strands)
    ph = rs.uniform(0, 2 * math.pi, n_strands)
    r = np.sqrt(1 - z * z)
    root = np.stack([r * np.cos(ph), z, r * np.sin(ph)], 1)
    length = rs.uniform(0.3, 0.6, n_strands)
    t = np.linspace(0.0, 1.0, n_cp)[None, :, None]
    nrm = root[:, None, :]
    grav = np.array([0.0, -1.0, 0.0])[None, None, :]
    jitter = rs.normal(0, 0.02, (n_strands, 1, 3))
    pts = nrm + nrm * (length[:, None, None] * t) * (1 - 0.5 * t) + grav * (length[:, None, None] * t * t * 0.9) + jitter * t
    rad = (4e-4 + (1e-4 - 4e-4) * t[..., 0]) * np.ones((n_strands, 1))
    # phantom points: first' = 2*p0 - p1, last' = 2*pn - pn-1 (BasisCurves.cpp:189-232); widths are already radii
    first = 2 * pts[:, :1] - pts[:, 1:2]
    last = 2 * pts[:, -1:] - pts[:, -2:-1]
    pts = np.concatenate([first, pts, last], 1)
    rad = np.concatenate([rad[:, :1], rad, rad[:, -1:]], 1)
    if n_prims <= 1:
        counts = np.full(n_strands, n_cp + 2, np.uint32)
        cid = sc.createCurve(counts, pts.reshape(-1, 3), rad.reshape(-1))
        sc.createInstance(S.INSTANCE_CURVE, cid, hair, np.eye(4))
    else:
        order = np.argsort(ph, kind="stable")
        for k, chunk in enumerate(np.array_split(order, n_prims)):
            moved = prim_offset != 0.0 and (n_moved is None or k < n_moved)
            counts = np.full(len(chunk), n_cp + 2, np.uint32)
            if moved and shared_xform:
                M = S.translate((prim_offset, 0.5 * prim_offset, 0.0)) @ S.rotate((0, 1, 0), 0.3)
                Mi = np.linalg.inv(M)
                p = pts[chunk].reshape(-1, 3)
                p = p @ Mi[:3, :3].T + Mi[:3, 3]
                cid = sc.createCurve(counts, p, rad[chunk].reshape(-1))
                sc.createInstance(S.INSTANCE_CURVE, cid, hair, M)
                continue
            off = np.array([(k + 1) * prim_offset if moved else 0.0, 0.0, 0.0])
            cid = sc.createCurve(counts, (pts[chunk] - off).reshape(-1, 3), rad[chunk].reshape(-1))
            sc.createInstance(S.INSTANCE_CURVE, cid, hair, S.translate(tuple(off)) if moved else np.eye(4))
    for pos, rot in [((2.5, 2.5, 2.5), (-45, 45, 0)), ((-3.0, 1.5, 1.0), (-20, -70, 0))]:
        xf = S.translate(pos) @ S.rotate((0, 1, 0), math.radians(rot[1])) @ S.rotate((1, 0, 0), math.radians(rot[0]))  # local -Z (the emitting side) towards the head
        sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 1.5, "height": 1.5, "color": (1, 1, 1),
                        "intensity": 25.0})
    cam = S.Camera(fov=40.0)
    cam.lookAt((0.0, 0.6, 4.2), (0.0, 0.2, 0.0))
    sc.addCamera(cam)
    return sc


def coffeemaker_standin(seed=1, target_tris=50000):
    """C1 "coffeemaker stand-in" (SURVEY 8d): lathe object ~50 k triangles on a 2-triangle ground plane, 1 rect
    light 0.4 x 0.4 intensity 160 (the commented default at HdStrelka/RenderPass.cpp:375-381)."""
    rs = np.random.RandomState(seed)
    sc = S.Scene()
    sc.addMaterial(S.MAT_DIFFUSE, (0.7, 0.7, 0.7))
    body = sc.addMaterial(S.MAT_PBR, (0.8, 0.3, 0.2), roughness=0.25, metallic=0.0)
    nu = int(math.sqrt(target_tris / 2.0 * 1.4))
    nv = int(target_tris / 2.0 / nu)
    pos, tris = _grid_mesh(_lathe_fn(rs), nu, nv)
    p = pos.astype(np.float64)
    if np.einsum("ij,ij->i", p[tris[:, 0]], np.cross(p[tris[:, 1]], p[tris[:, 2]])).sum() < 0:
        tris = tris[:, ::-1]
    m = _add_mesh(sc, pos, tris)
    g = _add_mesh(sc, *_quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4)))
    sc.createInstance(S.INSTANCE_MESH, g, 0, np.eye(4))
    sc.createInstance(S.INSTANCE_MESH, m, body, S.translate((0, 1.0, 0)))
    xf = S.translate((0.8, 3.0, 0.8)) @ S.rotate((1, 0, 0), math.radians(-90))
    sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 0.4, "height": 0.4, "color": (1, 1, 1),
                    "intensity": 160.0})
    cam = S.Camera(fov=45.0)
    cam.lookAt((2.6, 2.2, 3.2), (0.0, 0.9, 0.0))
    sc.addCamera(cam)
    return sc


def light_zoo(seed=11, with_rect=True):
    """Every light type of include/render/Lights.h in one small room: a SPHERE light (type 2: uniform sampling over the whole
    sphere with pdf 1/4pi, Lights.h:335-362; proxy = 16 x 16 UV sphere, scene.cpp:306-351), a DISK light (type 1: no sampler and
    no pdf in the reference -- it only shines when a path hits its 16-gon proxy, Lights.h:239-242), a rect and a distant light.
    A few objects of every material kind stand between them."""
    rs = np.random.RandomState(seed)
    sc = S.Scene()
    white = sc.addMaterial(S.MAT_DIFFUSE, (0.75, 0.75, 0.72))
    mats = [sc.addMaterial(S.MAT_DIFFUSE, (0.7, 0.3, 0.25)), sc.addMaterial(S.MAT_PBR, (0.3, 0.6, 0.3), roughness=0.3),
            sc.addMaterial(S.MAT_PBR, (0.9, 0.8, 0.4), roughness=0.2, metallic=1.0), sc.addMaterial(S.MAT_GLASS, (0.95, 0.97, 0.98), roughness=0.0, ior=1.5)]
    room = _add_mesh(sc, *_box_mesh((-2, 0, -2), (2, 3, 2), inward=True))
    sc.createInstance(S.INSTANCE_MESH, room, white, np.eye(4))
    meshes = []
    for k, fn in enumerate([_sphere_fn(rs, 0.08), _torus_fn(rs), _lathe_fn(rs), _sphere_fn(rs, 0.0)]):
        pos, tris = _grid_mesh(fn, 24, 16)
        p = pos.astype(np.float64)
        if np.einsum("ij,ij->i", p[tris[:, 0]], np.cross(p[tris[:, 1]], p[tris[:, 2]])).sum() < 0:
            tris = tris[:, ::-1]
        meshes.append(_add_mesh(sc, pos, tris))
    for k in range(8):
        x, z = -1.4 + 0.8 * (k % 4) + rs.uniform(-0.1, 0.1), (-0.6 if k < 4 else 0.7) + rs.uniform(-0.1, 0.1)
        s = rs.uniform(0.22, 0.3)
        xf = S.translate((x, 1.02 * s, z)) @ S.rotate((0, 1, 0), rs.uniform(0, 6.28)) @ S.scale((s, s, s))
        sc.createInstance(S.INSTANCE_MESH, meshes[k % 4], mats[(k + k // 4) % 4], xf)
    # (useXform: the proxy sphere is xform * scale(radius) and the sampler's centre xform * origin; without it the reference scales
    # the proxy by width / height instead of the radius, scene.h:331-343)
    sc.createLight({"type": 2, "xform": S.translate((-0.9, 2.2, 0.2)), "useXform": True, "radius": 0.25,
                    "color": (1.0, 0.85, 0.7), "intensity": 40.0})
    # disk normal = xform * (0,0,1) (scene.cpp:372-373); the light-hit test wants -dot(rayDir, normal) > 0 (OptixRender.cu:321-323):
    # local +Z turned to world -Y, the side the room sees
    xf = S.translate((1.0, 2.97, -0.3)) @ S.rotate((1, 0, 0), math.radians(90))
    sc.createLight({"type": 1, "xform": xf, "useXform": True, "radius": 0.45, "color": (0.8, 0.9, 1.0), "intensity": 30.0})
    if with_rect:
        xf = S.translate((0.0, 2.98, 1.2)) @ S.rotate((1, 0, 0), math.radians(-90))
        sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 0.8, "height": 0.5, "color": (1, 1, 1), "intensity": 12.0})
        xf = S.rotate((0, 1, 0), math.radians(20)) @ S.rotate((1, 0, 0), math.radians(-60))
        sc.createLight({"type": 3, "xform": xf, "useXform": True, "halfAngle": math.radians(3.0), "color": (1.0, 0.95, 0.85),
                        "intensity": 1.0, "radius": 0.0})
    cam = S.Camera(fov=55.0)
    cam.lookAt((0.0, 1.7, 1.95), (0.0, 1.0, 0.0))
    sc.addCamera(cam)
    return sc


def material_probe(kind, seed=5):
    """One material everywhere (floor, walls, three objects): a wrong branch of ONE BSDF cannot hide behind the others.
    kind: "diffuse" | "glossy" | "metal" | "glass" | "frosted" (OmniGlass with frosting_roughness, gltfloader.cpp:354-406)."""
    rs = np.random.RandomState(seed)
    sc = S.Scene()
    mat = {"diffuse": lambda: sc.addMaterial(S.MAT_DIFFUSE, (0.7, 0.55, 0.4)),
           "glossy": lambda: sc.addMaterial(S.MAT_PBR, (0.4, 0.6, 0.7), roughness=0.25, metallic=0.0),
           "metal": lambda: sc.addMaterial(S.MAT_PBR, (0.95, 0.75, 0.4), roughness=0.15, metallic=1.0),
           "glass": lambda: sc.addMaterial(S.MAT_GLASS, (0.9, 0.97, 0.95), roughness=0.0, ior=1.5),
           "frosted": lambda: sc.addMaterial(S.MAT_GLASS, (0.9, 0.97, 0.95), roughness=0.45, ior=1.5)}[kind]()
    room = _add_mesh(sc, *_box_mesh((-1.5, 0, -1.5), (1.5, 2.2, 1.5), inward=True))
    # a glass room would let nothing back: the shell of the two glass probes is diffuse white, the objects carry the material
    shell = mat if kind not in ("glass", "frosted") else sc.addMaterial(S.MAT_DIFFUSE, (0.75, 0.75, 0.75))
    sc.createInstance(S.INSTANCE_MESH, room, shell, np.eye(4))
    for k, fn in enumerate([_sphere_fn(rs, 0.0), _torus_fn(rs), _lathe_fn(rs)]):
        pos, tris = _grid_mesh(fn, 28, 18)
        p = pos.astype(np.float64)
        if np.einsum("ij,ij->i", p[tris[:, 0]], np.cross(p[tris[:, 1]], p[tris[:, 2]])).sum() < 0:
            tris = tris[:, ::-1]
        m = _add_mesh(sc, pos, tris)
        s = 0.33
        sc.createInstance(S.INSTANCE_MESH, m, mat, S.translate((-0.8 + 0.8 * k, 1.02 * s + 0.02, -0.1 * k)) @ S.rotate((0, 1, 0), 0.7 * k) @ S.scale((s, s, s)))
    xf = S.translate((0.0, 2.18, 0.3)) @ S.rotate((1, 0, 0), math.radians(-90))
    sc.createLight({"type": 0, "xform": xf, "useXform": True, "width": 0.9, "height": 0.6, "color": (1.0, 0.97, 0.9), "intensity": 18.0})
    sc.createLight({"type": 2, "xform": S.translate((-1.0, 1.6, 0.9)), "useXform": True, "radius": 0.12,
                    "color": (0.9, 0.95, 1.0), "intensity": 60.0})
    cam = S.Camera(fov=50.0)
    cam.lookAt((0.0, 1.1, 1.45), (0.0, 0.55, -0.2))
    sc.addCamera(cam)
    return sc


def random_rays(n, seed, lo, hi, tmax=1e16):
    """Uniform random origins in [lo,hi]^3 and uniform directions, as skh_ray records."""
    rs = np.random.RandomState(seed)
    rays = np.zeros(n, S.RAY)
    rays["origin"] = rs.uniform(lo, hi, (n, 3))
    d = rs.normal(size=(n, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmin"] = 0.0
    rays["tmax"] = tmax
    return rays
