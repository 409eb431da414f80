"""Scene compiler: a tree of shape objects -> the flat fp64 arrays of `rtc_scene_desc`.

Input is duck-typed: this package's `shapes`/`renderer` objects or the reference's own
(`World.contains`, `.transform.arr`, `.material.*`, `.children`, `.left/.right/.op`,
`.minimum/.maximum/.closed`, `.p1..p3`, `.n1..n3`; classes are told apart by class name).

What it does, once per scene, on the host:
  * numbers the primitive shapes ("leaves") depth-first from `world.contains`;
  * collapses each leaf's chain of per-level inverses (reference shapes.py:271-273 applied at
    every tree level, :282-286) into ONE world->object 3x4 -- its transpose is the collapsed
    `normal_to_world` chain (:288-296), since per-level renormalisation only rescales;
  * moves triangles to world space (points by object->world, normals by the transposed
    world->object) and rescales the Moller-Trumbore determinant epsilon by |det(object->world)|
    so the `abs(det) < EPSILON` rejection (:815, :863) is the same test;
  * computes a world-space AABB per leaf / per top-most CSG (the library builds its BVH over
    them); planes and other unbounded things go to a list tested for every ray;
  * records every top-most CSG as a post-ordered list of nodes with contiguous leaf ranges,
    each node with its reference `bounds()` box (:391-396) and world->csg-local matrix;
  * turns each material's pattern tree into a postfix program over pattern nodes;
  * keeps a mesh the loader made more than once (the same block of vertex data under several
    transforms: bounding_boxes.yml loads dragon.obj six times) ONCE, in its object space, with an
    instance per use (rtc_scene_desc::inst) instead of flattening every copy to world space.
"""
import hashlib
import os

import numpy as np

EPSILON = 0.000001

K_SPHERE, K_PLANE, K_CUBE, K_CYLINDER, K_CONE, K_TRIANGLE, K_CSG_ROOT, K_INSTANCE = 0, 1, 2, 3, 4, 5, 8, 9
MAX_INSTANCES = 8            # RTC_MAX_INSTANCES (include/rtc_api.h)
MIN_INSTANCE_TRIS = 1024     # smaller repeated meshes are flattened like everything else
# instancing="auto": repeated meshes are instanced when that keeps at least this many triangles off the GPU (~ 290 bytes each in
# the flattened form: rows in both precisions, normals, nodes, per-leaf entries).  Below it the flattened form is kept because it
# is the faster one in fp64 (dragons 1080p: 0.43 flattened, 0.47 ms instanced -- the object-space ray beside the world-space one
# costs the fp64 kernel registers it does not have; fp32: 0.415 -> 0.394 ms), DESIGN.md.
AUTO_INSTANCE_SAVED_TRIS = 250000
_KIND = {"Sphere": K_SPHERE, "Plane": K_PLANE, "Cube": K_CUBE, "Cylinder": K_CYLINDER, "Cone": K_CONE,
         "Triangle": K_TRIANGLE, "SmoothTriangle": K_TRIANGLE}
_CSG_OP = {"CSGUnion": 0, "CSGIntersect": 1, "CSGDifference": 2}
_PAT_SIMPLE = {"Pattern": 0, "Checkers": 1, "Gradient": 2, "RadialGradient": 3, "Ring": 4, "Stripe": 5}
_TWO_COLOR = ("Checkers", "Gradient", "RadialGradient", "Ring", "Stripe")
_UV = {"UVChecker": 20, "UVAlignCheck": 21, "UVTexturePattern": 22}
_UVMAP = {"PlaneUVMap": 9, "SphereUVMap": 10}
OP_EVAL, OP_EVAL_AB, OP_BLEND, OP_PERTURB, OP_RESTORE = 0, 1, 2, 3, 4
PAT_PERTURBED = 8
_MAX_HITS = {K_SPHERE: 2, K_PLANE: 1, K_CUBE: 2, K_CYLINDER: 4, K_CONE: 4, K_TRIANGLE: 1}


def prim_code(kind, noshadow, index):
    return np.int32(np.uint32((kind << 28) | ((1 if noshadow else 0) << 27) | (index & 0x07FFFFFF)).view(np.int32))


def _a(x, n=None):
    a = np.asarray(getattr(x, "arr", x), dtype=np.float64)
    return a if n is None else a[:n]


def _cname(x):
    return x.__name__ if isinstance(x, type) else type(x).__name__


class CompiledScene(object):
    """`arrays`: name -> contiguous ndarray, named as the fields of `rtc_scene_desc`;
    `counts`: the n_* fields; `leaf_objects`: the shape object behind each leaf id."""

    def __init__(self):
        self.arrays = {}
        self.counts = {}
        self.leaf_objects = []


class _Compiler(object):
    def __init__(self, instancing="auto"):
        env = os.environ.get("RTC_INSTANCING")               # development: 1 / 0 overrides the caller's choice
        self.instancing = (env == "1") if env in ("0", "1") else instancing
        self.leaf_kind, self.leaf_xform, self.leaf_material, self.leaf_tri = [], [], [], []
        self.leaf_minmax, self.leaf_closed, self.leaf_objects = [], [], []
        self.leaf_lo = []                            # (leaf, lo, hi): world AABB of the analytic leaves that have a finite one
        self.xforms, self._xform_of = [], {}
        self.tri_rows, self.tri_xform, self.tri_leaf, self.tri_flat = [], [], [], []
        self.tri_single = []                         # which triangle rows came one by one (tri_rows, tri_xform, tri_leaf, tri_flat hold theirs, in this order)
        self.leaf_single = []                        # likewise for leaves: the leaf ids the per-leaf lists above are for
        self.runs = []                               # (first leaf, count, xform row, material, first triangle row, (n, 18) array, in_csg): loader-made runs, flattened in bulk
        self.n_leaves = self.n_tris = 0
        self.items, self.unbounded = [], []          # (kind, leaf or csg root)
        self.csg_roots, self.csg_nodes, self.csg_bbox = [], [], []
        self.materials, self._mat_of = [], {}
        self.pat_nodes, self._pat_of = [], {}
        self.programs, self.instr = [], []
        self.textures, self._tex_of = [], {}
        self._inv_cache = {}
        self._ident = np.eye(4)
        self._ident_key = self._ident.tobytes()
        self._xform_by_id = {}                       # id(matrix) -> (matrix kept alive, row)

    # ---- transforms ------------------------------------------------------------------
    def _inverse(self, t):
        arr = _a(t)
        key = arr.tobytes()
        if key == self._ident_key:
            return self._ident            # one shared object: lets shape() skip the product (a mesh's triangles)
        inv = self._inv_cache.get(key)
        if inv is None:
            inv = np.linalg.inv(arr)      # matrix.py:393
            self._inv_cache[key] = inv
        return inv

    def _xform(self, m):
        hit = self._xform_by_id.get(id(m))
        if hit is not None:
            return hit[1]
        key = m.tobytes()
        idx = self._xform_of.get(key)
        if idx is None:
            idx = self._xform_of[key] = len(self.xforms)
            self.xforms.append(m)
        self._xform_by_id[id(m)] = (m, idx)
        return idx

    # ---- tree walk -------------------------------------------------------------------
    def shape(self, obj, m_parent, in_csg):
        name = type(obj).__name__
        inv = self._inverse(obj.transform)
        m = inv if m_parent is None else (m_parent if inv is self._ident else inv @ m_parent)
        if name == "Group":
            kids = list(obj.children)
            at = 0
            for run in sorted(getattr(obj, "triangle_runs", None) or [], key=lambda r: r.start):
                if run.start < at or not run.still_valid(kids):
                    continue                                   # not what the loader made any more: the objects are read one by one
                for c in kids[at:run.start]:
                    self.shape(c, m, in_csg)
                self._triangle_run(run, kids, m, in_csg)
                at = run.start + run.count
            for c in kids[at:]:
                self.shape(c, m, in_csg)
        elif name == "CSG":
            if in_csg:
                self._csg_node(obj, m)
            else:
                first_node, first_leaf = len(self.csg_nodes), self.n_leaves
                self._csg_node(obj, m)
                self.csg_roots.append([first_node, len(self.csg_nodes) - first_node, first_leaf, self.n_leaves])
        elif name in _KIND:
            self._leaf(obj, name, m, in_csg)
        else:
            raise ValueError("unknown shape class " + name)

    def _csg_node(self, obj, m):
        l0 = self.n_leaves
        self.shape(obj.left, m, True)
        l1 = self.n_leaves
        self.shape(obj.right, m, True)
        l2 = self.n_leaves
        b = obj.bounds()
        self.csg_nodes.append([_CSG_OP[_cname(obj.op)], l0, l1, l2, self._xform(m)])
        self.csg_bbox.append(list(_a(b.min, 3)) + list(_a(b.max, 3)))

    def _triangle_run(self, run, kids, m, in_csg):
        """run.count SmoothTriangle leaves (identity transforms, one material) from the loader's own block of vertex data."""
        n = run.count
        self.runs.append((self.n_leaves, n, self._xform(m), self._material(run.material), self.n_tris, run.rows, in_csg))
        self.leaf_objects.extend(kids[run.start:run.start + n])
        self.n_leaves += n
        self.n_tris += n

    def _leaf(self, obj, name, m, in_csg):
        kind = _KIND[name]
        leaf = self.n_leaves
        self.n_leaves += 1
        self.leaf_single.append(leaf)
        mat = self._material(obj.material)
        self.leaf_kind.append(kind)
        self.leaf_xform.append(self._xform(m))
        self.leaf_material.append(mat)
        self.leaf_objects.append(obj)
        lo, hi, minmax, closed = None, None, (0.0, 0.0), 0
        if kind == K_TRIANGLE:
            self.leaf_tri.append(self.n_tris)
            flat = name == "Triangle"
            # the raw vertex (and normal) arrays; stacked in bulk by finish()
            p1, p2, p3 = obj.p1, obj.p2, obj.p3
            if flat:
                self.tri_rows.append((getattr(p1, "arr", p1), getattr(p2, "arr", p2), getattr(p3, "arr", p3)))
            else:
                n1, n2, n3 = obj.n1, obj.n2, obj.n3
                self.tri_rows.append((getattr(p1, "arr", p1), getattr(p2, "arr", p2), getattr(p3, "arr", p3),
                                      getattr(n1, "arr", n1), getattr(n2, "arr", n2), getattr(n3, "arr", n3)))
            self.tri_single.append(self.n_tris)
            self.n_tris += 1
            self.tri_xform.append(self.leaf_xform[-1])
            self.tri_leaf.append(leaf)
            self.tri_flat.append(flat)
        else:
            self.leaf_tri.append(-1)
            if kind in (K_CYLINDER, K_CONE):
                minmax = (float(obj.minimum), float(obj.maximum))
                closed = 1 if obj.closed else 0
            lo, hi = self._analytic_bounds(kind, minmax, m, closed)
        self.leaf_minmax.append(minmax)
        self.leaf_closed.append(closed)
        if lo is not None:
            self.leaf_lo.append((leaf, lo, hi))        # analytic leaves with a finite box (triangles get theirs in bulk, finish())
        if not in_csg:
            (self.items if (kind == K_TRIANGLE or lo is not None) else self.unbounded).append((kind, leaf))

    @staticmethod
    def _analytic_bounds(kind, minmax, m, closed=0):
        """World AABB of what the leaf's local_intersect can report (shapes.py:298-300, 561-562, 708-713, 795-796).
        For a closed cone that is MORE than the reference's own bounds(): its cap test accepts x^2 + z^2 <= |y|
        (shapes.py:689-693: |y|, not y^2), a disc of radius sqrt|y| -- a brim around the cone wherever |y| < 1 -- and
        the reference, which never culls, renders it; a box without the brim would cull those hits away."""
        if kind == K_PLANE:
            return None, None
        if kind in (K_SPHERE, K_CUBE):
            l, h = np.array([-1.0, -1, -1]), np.array([1.0, 1, 1])
        else:
            if not (np.isfinite(minmax[0]) and np.isfinite(minmax[1])):
                return None, None
            r = 1.0 if kind == K_CYLINDER else max(abs(minmax[0]), abs(minmax[1]))
            if kind == K_CONE and closed:
                r = max(r, np.sqrt(r))
            l, h = np.array([-r, minmax[0], -r]), np.array([r, minmax[1], r])
        a = np.linalg.inv(m)
        corners = np.array([[x, y, z, 1.0] for x in (l[0], h[0]) for y in (l[1], h[1]) for z in (l[2], h[2])])
        w = corners @ a.T
        return w[:, :3].min(axis=0), w[:, :3].max(axis=0)

    # ---- materials and patterns --------------------------------------------------------
    def _material(self, m):
        idx = self._mat_of.get(id(m))
        if idx is None:
            idx = self._mat_of[id(m)] = len(self.materials)
            prog = -1 if m.pattern is None else self._program(m.pattern)
            self.materials.append((list(_a(m.color, 3)) + [float(m.ambient), float(m.diffuse), float(m.specular),
                                   float(m.shininess), float(m.reflective), float(m.transparency),
                                   float(m.refractive_index)], 1 if m.casts_shadow else 0, prog, m))
        return idx

    def _program(self, p):
        code = []
        self._emit(p, code)
        self.programs.append((len(self.instr), len(code)))
        self.instr.extend(code)
        return len(self.programs) - 1

    def _emit(self, p, code):
        name = type(p).__name__
        if name == "BlendedPattern":                     # shapes.py:1874-1877
            self._emit(p.p1, code)
            self._emit(p.p2, code)
            code.append((OP_BLEND, 0))
        elif name == "NestedPattern":                    # shapes.py:1892-1895
            if p.p2 is p.p1 or p.p3 is p.p1:
                raise NotImplementedError("NestedPattern whose primary is also its own sub-pattern is "
                                          "stateful in the reference (it overwrites p1.a/p1.b while evaluating p1)")
            if type(p.p1).__name__ in _TWO_COLOR:
                self._emit(p.p2, code)
                self._emit(p.p3, code)
                code.append((OP_EVAL_AB, self._pat_node(p.p1)))
            else:
                # the colours written into p1.a / p1.b are never read by this kind of primary
                self._emit(p.p1, code)
        elif name == "PerturbedPattern":                 # shapes.py:1911-1925 (noise: parity unpinned)
            if not (float(p.freq) != 0.0 and int(p.octaves) == p.octaves and 1 <= int(p.octaves) <= 16):
                raise ValueError("PerturbedPattern: frequency must not be 0 and octaves must be an integer in 1..16 "
                                 "(noise.pnoise3 raises for octaves < 1)")
            code.append((OP_PERTURB, self._pat_node(p)))
            self._emit(p.p1, code)
            code.append((OP_RESTORE, 0))
        else:
            code.append((OP_EVAL, self._pat_node(p)))

    def _pat_node(self, p):
        idx = self._pat_of.get(id(p))
        if idx is not None:
            return idx
        idx = self._pat_of[id(p)] = len(self.pat_nodes)
        name = type(p).__name__
        rec = {"type": None, "inv": self._inverse(p.transform), "ab": list(_a(p.a, 3)) + list(_a(p.b, 3)),
               "child": [-1] * 6, "param": [0.0] * 4, "colors": [0.0] * 15, "tex": -1}
        self.pat_nodes.append(rec)
        if name in _PAT_SIMPLE:
            rec["type"] = _PAT_SIMPLE[name]
        elif name == "PerturbedPattern":
            rec["type"] = PAT_PERTURBED
            rec["param"][:3] = [float(p.freq), float(p.scale_factor), float(int(p.octaves))]
        elif name in _UV:
            rec["type"] = _UV[name]
            if name == "UVChecker":
                rec["param"][:2] = [float(p.width), float(p.height)]
            elif name == "UVAlignCheck":
                rec["colors"] = [float(x) for c in (p.main, p.ul, p.ur, p.bl, p.br) for x in c]
            else:
                rec["tex"] = self._texture(p.canvas)
        elif name == "TextureMapPattern":
            rec["type"] = _UVMAP[_cname(p.uv_map)]
            rec["child"][0] = self._pat_node(p.uv_pattern)
        elif name == "CylinderMapPattern":
            rec["type"] = 11
            rec["child"][:2] = [self._pat_node(f) for f in p.faces]
        elif name == "CubeMapPattern":
            rec["type"] = 12
            rec["child"][:6] = [self._pat_node(f) for f in p.faces]
        else:
            raise ValueError("unknown pattern class " + name)
        return idx

    def _texture(self, tc):
        idx = self._tex_of.get(id(tc))
        if idx is None:
            px = np.asarray(tc.pixels)
            if px.ndim != 3 or px.shape[2] < 3:
                raise ValueError("image textures must be RGB(A); the reference indexes three channels (canvas.py:52-54)")
            idx = self._tex_of[id(tc)] = len(self.textures)
            self.textures.append((np.ascontiguousarray(px[:, :, :3].astype(np.uint8)), float(np.max(px))))
        return idx

    # ---- assembly ------------------------------------------------------------------------
    def finish(self, lights):
        f64, i32 = np.float64, np.int32
        cs = CompiledScene()
        A, C = cs.arrays, cs.counts
        n_leaves, n_tris = self.n_leaves, self.n_tris
        xf = np.array(self.xforms, f64).reshape(-1, 4, 4) if self.xforms else np.zeros((0, 4, 4))
        mat_shadow = np.array([m[1] for m in self.materials] or [1], i32)

        # ---- repeated meshes: which loader-made runs become instances of one object-space mesh ----
        mesh_of_run, meshes = {}, []                       # run index -> (mesh, instance); meshes: (first mesh row, block)
        if self.instancing and not self.csg_roots and not any(op == OP_PERTURB for op, _ in self.instr):    # (those scenes' kernels carry no instance walk)
            groups, by_size = {}, {}
            for leaf0, n, xrow, mat, tri0, block, in_csg in self.runs:
                if not in_csg and n >= MIN_INSTANCE_TRIS:
                    by_size[n] = by_size.get(n, 0) + 1
            # (hashing the vertex blocks costs milliseconds: not when even equal sizes cannot add up to what "auto" asks for)
            worth = self.instancing is True or sum(n * (c - 1) for n, c in by_size.items()) >= AUTO_INSTANCE_SAVED_TRIS
            for ri, (leaf0, n, xrow, mat, tri0, block, in_csg) in enumerate(self.runs if worth else []):
                if not in_csg and n >= MIN_INSTANCE_TRIS and by_size[n] >= 2:
                    groups.setdefault((n, hashlib.sha1(np.ascontiguousarray(block).tobytes()).digest()), []).append(ri)
            n_inst, row0 = 0, 0
            picked = sorted(groups.items(), key=lambda kv: -kv[0][0] * len(kv[1]))                  # the most triangles saved first
            if self.instancing == "auto" and sum(k[0] * (min(len(m), MAX_INSTANCES) - 1) for k, m in picked if len(m) >= 2) < AUTO_INSTANCE_SAVED_TRIS:
                picked = []
            for key, members in picked:
                members = members[:MAX_INSTANCES - n_inst]
                if len(members) < 2:
                    continue
                for ri in members:
                    mesh_of_run[ri] = (len(meshes), n_inst)
                    n_inst += 1
                meshes.append((row0, self.runs[members[0]][5]))
                row0 += key[0]
        if mesh_of_run:
            # their triangles leave the world-space triangle table: the rows after them move up
            gone = np.zeros(n_tris + 1, np.int64)
            for ri in mesh_of_run:
                _, n, _, _, tri0, _, _ = self.runs[ri]
                gone[tri0:tri0 + n] = 1
            new_row = np.arange(n_tris + 1) - np.concatenate([[0], np.cumsum(gone)[:-1]])
            self.tri_single = [int(new_row[t]) for t in self.tri_single]
            self.leaf_tri = [int(new_row[t]) if t >= 0 else t for t in self.leaf_tri]
            self.runs = [(l0, n, x, m, int(new_row[t0]), b, c) for (l0, n, x, m, t0, b, c) in self.runs]
            n_tris -= int(gone.sum())

        # per-leaf and per-triangle tables: the leaves that were walked one by one (the lists) and the loader-made runs (slices)
        ls, ts = np.array(self.leaf_single, np.int64), np.array(self.tri_single, np.int64)

        def table(n, dt, at, singles, width=None):
            out = np.zeros(n if width is None else (n, width), dt)
            if len(at):
                out[at] = np.array(singles, dt).reshape((-1,) if width is None else (-1, width))
            return out
        leaf_kind, leaf_xform = table(n_leaves, i32, ls, self.leaf_kind), table(n_leaves, i32, ls, self.leaf_xform)
        leaf_material, leaf_tri = table(n_leaves, i32, ls, self.leaf_material), table(n_leaves, i32, ls, self.leaf_tri)
        leaf_minmax, leaf_closed = table(n_leaves, f64, ls, self.leaf_minmax, 2), table(n_leaves, i32, ls, self.leaf_closed)
        txf, tri_leaf, flat = table(n_tris, i32, ts, self.tri_xform), table(n_tris, i32, ts, self.tri_leaf), table(n_tris, bool, ts, self.tri_flat)
        rows = np.zeros((n_tris, 18))
        run_items = []
        inst_rows, inst_prim, inst_lo, inst_hi = [None] * len(mesh_of_run), [None] * len(mesh_of_run), [None] * len(mesh_of_run), [None] * len(mesh_of_run)
        for ri, (leaf0, n, xrow, mat, tri0, block, in_csg) in enumerate(self.runs):
            leaf_kind[leaf0:leaf0 + n], leaf_xform[leaf0:leaf0 + n], leaf_material[leaf0:leaf0 + n] = K_TRIANGLE, xrow, mat
            if ri in mesh_of_run:
                mesh, q = mesh_of_run[ri]
                leaf_tri[leaf0:leaf0 + n] = -2 - (meshes[mesh][0] + np.arange(n))
                a = np.linalg.inv(xf[xrow])                 # object -> world: the instance's box, from its own vertices
                pts = block[:, :9].reshape(-1, 3) @ a[:3, :3].T + a[:3, 3]
                inst_rows[q] = [mesh, xrow, leaf0, mat]
                inst_prim[q] = prim_code(K_INSTANCE, mat_shadow[mat] == 0, q)
                inst_lo[q], inst_hi[q] = pts.min(axis=0), pts.max(axis=0)
                continue
            leaf_tri[leaf0:leaf0 + n] = np.arange(tri0, tri0 + n)
            txf[tri0:tri0 + n], tri_leaf[tri0:tri0 + n] = xrow, np.arange(leaf0, leaf0 + n)
            if not in_csg:
                run_items.append(np.stack([np.full(n, K_TRIANGLE, np.int64), np.arange(leaf0, leaf0 + n, dtype=np.int64)], axis=1))
        leaf_noshadow = (mat_shadow[leaf_material] == 0) if n_leaves else np.zeros(0, bool)

        # triangles -> world space, in bulk per transform
        tri_geom = np.zeros((max(n_tris, 1), 12))
        tri_norm = np.zeros((max(n_tris, 1), 12))
        tri_lo = np.zeros((n_tris, 3))
        tri_hi = np.zeros((n_tris, 3))
        if n_tris:
            single = ts                                                                 # rows that came object by object
            for is_flat, k in ((True, 3), (False, 6)):
                pick = np.nonzero(flat[single] == is_flat)[0] if single.size else single
                if pick.size:
                    try:
                        block = np.array([self.tri_rows[i] for i in pick], f64)          # (n, k, 3 or 4)
                    except ValueError:                                                  # mixed 3- and 4-component tuples
                        block = np.array([[np.asarray(v, f64)[:3] for v in self.tri_rows[i]] for i in pick], f64)
                    rows[single[pick], :3 * k] = block[:, :, :3].reshape(pick.size, 3 * k)
            def to_world(r, fl, xi, where):
                """rows r (n, 18: p1 p2 p3 n1 n2 n3, object space) under transform row xi -> tri_geom / tri_norm / boxes at `where`
                (a slice for a loader-made run, an index array for triangles that came one by one)"""
                m = xf[xi]
                a = np.linalg.inv(m)                       # object -> world
                a3, t3, m3 = a[:3, :3], a[:3, 3], m[:3, :3]
                p = [r[:, 3 * k:3 * k + 3] @ a3.T + t3 for k in range(3)]
                if fl is None:                             # a loader's run: smooth triangles, the normals are in the rows
                    normals = [r[:, 9 + 3 * k:12 + 3 * k] @ m3 for k in range(3)]                              # M^T n
                else:
                    e1o, e2o = r[:, 3:6] - r[:, 0:3], r[:, 6:9] - r[:, 0:3]
                    fn = np.cross(e2o, e1o)                # shapes.py:807
                    with np.errstate(invalid="ignore", divide="ignore"):
                        fn = fn * (1.0 / np.sqrt(np.sum(fn * fn, axis=1, keepdims=True)))
                    normals = [np.where(fl[:, None], fn, r[:, 9 + 3 * k:12 + 3 * k]) @ m3 for k in range(3)]
                g = np.empty((len(r), 12))
                g[:, 0:3], g[:, 3:6], g[:, 6:9] = p[0], p[1] - p[0], p[2] - p[0]
                g[:, 9], g[:, 10:12] = EPSILON * abs(np.linalg.det(a3)), 0.0
                tri_geom[where] = g
                nn = np.zeros((len(r), 12))
                nn[:, 0:3], nn[:, 3:6], nn[:, 6:9] = normals
                tri_norm[where] = nn
                tri_lo[where] = np.minimum(np.minimum(p[0], p[1]), p[2])
                tri_hi[where] = np.maximum(np.maximum(p[0], p[1]), p[2])
            for ri, (leaf0, n, xrow, mat, tri0, block, in_csg) in enumerate(self.runs):
                if ri not in mesh_of_run:
                    to_world(block, None, xrow, slice(tri0, tri0 + n))
            if single.size:
                for xi in np.unique(txf[single]):
                    sel = single[txf[single] == xi]
                    to_world(rows[sel], flat[sel], xi, sel)
        # world AABB per leaf as arrays (NaN = no finite box): analytic leaves from the walk, triangles in bulk
        leaf_lo = np.full((n_leaves, 3), np.nan)
        leaf_hi = np.full((n_leaves, 3), np.nan)
        for leaf, lo, hi in self.leaf_lo:
            leaf_lo[leaf], leaf_hi[leaf] = lo, hi
        if n_tris:
            leaf_lo[tri_leaf], leaf_hi[tri_leaf] = tri_lo, tri_hi

        # items for the BVH / the unbounded list (in leaf order, as one walk of the tree would have listed them)

        def codes(pairs, more=()):
            both = np.concatenate([np.array(pairs, np.int64).reshape(-1, 2)] + list(more))
            if len(more):
                both = both[np.argsort(both[:, 1], kind="stable")]
            if not len(both):
                return np.zeros(0, i32), np.zeros(0, np.int64)
            kinds, leaves = both[:, 0].astype(np.uint32), both[:, 1]
            idx = np.where(kinds == K_TRIANGLE, leaf_tri[leaves], leaves).astype(np.uint32)
            return ((kinds << 28) | (leaf_noshadow[leaves].astype(np.uint32) << 27) | (idx & 0x07FFFFFF)).view(i32), leaves

        item_prim, item_leaves = codes(self.items, run_items)
        item_lo, item_hi = leaf_lo[item_leaves], leaf_hi[item_leaves]
        unb = list(codes(self.unbounded)[0])
        root_prim, root_lo, root_hi = [], [], []
        for root, (_, _, l0, l2) in enumerate(self.csg_roots):
            los, his = leaf_lo[l0:l2], leaf_hi[l0:l2]
            if not (np.all(np.isfinite(los)) and np.all(np.isfinite(his))):
                unb.append(prim_code(K_CSG_ROOT, False, root))
            else:
                root_prim.append(prim_code(K_CSG_ROOT, False, root))
                root_lo.append(np.min(los, axis=0))
                root_hi.append(np.max(his, axis=0))
        root_prim, root_lo, root_hi = root_prim + inst_prim, root_lo + inst_lo, root_hi + inst_hi
        if root_prim:
            item_prim = np.concatenate([item_prim, np.array(root_prim, i32)])
            item_lo = np.concatenate([item_lo, np.array(root_lo, f64).reshape(-1, 3)])
            item_hi = np.concatenate([item_hi, np.array(root_hi, f64).reshape(-1, 3)])
        # an item with non-finite bounds (degenerate transform) is tested for every ray instead
        if len(item_prim):
            ok = np.all(np.isfinite(item_lo), axis=1) & np.all(np.isfinite(item_hi), axis=1)
            if not ok.all():
                unb.extend(item_prim[~ok])
                item_prim, item_lo, item_hi = item_prim[ok], item_lo[ok], item_hi[ok]

        def arr(x, dt, shape):
            a = np.ascontiguousarray(np.array(x, dt).reshape(shape))
            return a if a.size else np.zeros([max(1, s) if s != -1 else 1 for s in shape], dt)

        A["leaf_kind"], A["leaf_xform"], A["leaf_material"] = arr(leaf_kind, i32, (-1,)), arr(leaf_xform, i32, (-1,)), arr(leaf_material, i32, (-1,))
        A["leaf_tri"], A["leaf_minmax"], A["leaf_closed"] = arr(leaf_tri, i32, (-1,)), arr(leaf_minmax, f64, (-1, 2)), arr(leaf_closed, i32, (-1,))
        A["xform"] = arr(xf[:, :3, :], f64, (-1, 12))
        A["tri_geom"] = np.ascontiguousarray(tri_geom)
        A["tri_norm"] = np.ascontiguousarray(tri_norm)
        A["tri_leaf"] = arr(tri_leaf, i32, (-1,))
        n_mesh_tris = sum(len(b) for _, b in meshes)
        mesh_geom, mesh_norm = np.zeros((max(n_mesh_tris, 1), 12)), np.zeros((max(n_mesh_tris, 1), 12))
        for r0, b in meshes:                               # object space, as the reference's own triangles hold it (shapes.py:799-811, 846-858)
            g = mesh_geom[r0:r0 + len(b)]
            g[:, 0:3], g[:, 3:6], g[:, 6:9], g[:, 9] = b[:, 0:3], b[:, 3:6] - b[:, 0:3], b[:, 6:9] - b[:, 0:3], EPSILON
            mesh_norm[r0:r0 + len(b), 0:9] = b[:, 9:18]
        A["mesh_geom"], A["mesh_norm"] = np.ascontiguousarray(mesh_geom), np.ascontiguousarray(mesh_norm)
        A["mesh_range"] = arr([[r0, len(b)] for r0, b in meshes], i32, (-1, 2))
        A["inst"] = arr(inst_rows, i32, (-1, 4))
        A["item_prim"] = arr(item_prim, i32, (-1,))
        A["item_lo"] = arr(item_lo, f64, (-1, 3))
        A["item_hi"] = arr(item_hi, f64, (-1, 3))
        A["unbounded_prim"] = arr(unb, i32, (-1,))
        A["csg_root"] = arr(self.csg_roots, i32, (-1, 4))
        A["csg_node"] = arr(self.csg_nodes, i32, (-1, 5))
        A["csg_bbox"] = arr(self.csg_bbox, f64, (-1, 6))
        A["mat"] = arr([m[0] for m in self.materials], f64, (-1, 10))
        A["mat_shadow"] = arr([m[1] for m in self.materials], i32, (-1,))
        A["mat_program"] = arr([m[2] for m in self.materials], i32, (-1,))
        P = self.pat_nodes
        A["pat_type"] = arr([p["type"] for p in P], i32, (-1,))
        A["pat_inv"] = arr([p["inv"][:3, :] for p in P], f64, (-1, 12))
        A["pat_ab"] = arr([p["ab"] for p in P], f64, (-1, 6))
        A["pat_child"] = arr([p["child"] for p in P], i32, (-1, 6))
        A["pat_param"] = arr([p["param"] for p in P], f64, (-1, 4))
        A["pat_colors"] = arr([p["colors"] for p in P], f64, (-1, 15))
        A["pat_tex"] = arr([p["tex"] for p in P], i32, (-1,))
        A["prog_off"] = arr([p[0] for p in self.programs], i32, (-1,))
        A["prog_len"] = arr([p[1] for p in self.programs], i32, (-1,))
        A["instr"] = arr(self.instr, i32, (-1, 2))
        offs, off = [], 0
        for px, _ in self.textures:
            offs.append(off)
            off += px.size
        A["tex_off"] = arr(offs, np.int64, (-1,))
        A["tex_w"] = arr([t[0].shape[1] for t in self.textures], i32, (-1,))
        A["tex_h"] = arr([t[0].shape[0] for t in self.textures], i32, (-1,))
        A["tex_scale"] = arr([t[1] for t in self.textures], f64, (-1,))
        A["tex_data"] = (np.concatenate([t[0].ravel() for t in self.textures]) if self.textures else np.zeros(1, np.uint8))
        typ, geom, steps, inten = [], [], [], []
        for l in lights:
            if hasattr(l, "position"):
                typ.append(0)
                geom.append(list(_a(l.position, 3)) + [0.0] * 6)
                steps.append([1, 1, 0])
            else:
                typ.append(1)
                geom.append(list(_a(l.corner, 3)) + list(_a(l.uvec, 3)) + list(_a(l.vvec, 3)))
                steps.append([int(l.usteps), int(l.vsteps), 1 if l.jitter else 0])
            inten.append(list(_a(l.intensity, 3)))
        A["light_type"] = arr(typ, i32, (-1,))
        A["light_geom"] = arr(geom, f64, (-1, 9))
        A["light_steps"] = arr(steps, i32, (-1, 3))
        A["light_intensity"] = arr(inten, f64, (-1, 3))
        C.update(n_leaves=n_leaves, n_xforms=len(self.xforms), n_tris=n_tris, n_items=int(len(item_prim)),
                 n_unbounded=len(unb), n_csg_roots=len(self.csg_roots), n_csg_nodes=len(self.csg_nodes),
                 n_materials=len(self.materials), n_pat_nodes=len(P), n_programs=len(self.programs),
                 n_instr=len(self.instr), n_textures=len(self.textures), tex_bytes=off, n_lights=len(lights),
                 n_mesh_tris=n_mesh_tris, n_meshes=len(meshes), n_instances=len(inst_rows))
        cs.leaf_objects = self.leaf_objects
        return cs


def compile_scene(world, instancing="auto"):
    """World (lights + contains) -> CompiledScene.  instancing: True = meshes the loader made more than once are kept once, in
    object space, with an instance per use (a third of the device memory on the dragons; faster in fp32, slower in fp64);
    False = every copy flattened to world space (what the triangle-update entry points replace rows of); "auto" (default) =
    instanced only when that keeps AUTO_INSTANCE_SAVED_TRIS triangles off the GPU."""
    c = _Compiler(instancing)
    for obj in world.contains:
        c.shape(obj, None, False)
    return c.finish(list(world.lights or []))
