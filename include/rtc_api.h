/* rtc_api.h -- C ABI of librtc_hip.so, the MI355X trace loop behind
 * gbordelon/ray_tracer_challenge's `render(camera, world) -> canvas`.
 *
 * The reference is pure Python and has no FFI of its own; the functions below are what a
 * binding for its hot path -- render / render_multi (renderer.py:753-835) and everything they
 * call per pixel (ray_for_pixel :715-751, color_at :562-611, intersect_world :67-79,
 * prepare_computations :98-244, shade_hit :408-526, lighting :990-1090, is_shadowed :528-560,
 * reflected_color :355-406, refracted_color :285-353, schlick :246-283, and the per-ray parts
 * of shapes.py) -- has to call.  INTEGRATION.md shows the ctypes stub for the reference side.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 or a negative
 * rtc_status and leaves a message for rtc_last_error() (thread-local); nothing throws or exits
 * across the ABI; output buffers are caller-allocated and written only on success.
 * All scene inputs are fp64 (the reference's dtype); the library keeps fp64 and fp32 device
 * copies.  A scene handle is immutable after creation and belongs to one GPU.
 */
#ifndef RTC_API_H
#define RTC_API_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTC_VERSION 100

typedef enum {
    RTC_OK = 0,
    RTC_ERR_INVALID = -1,   /* bad argument / inconsistent scene description */
    RTC_ERR_NO_DEVICE = -2, /* no usable GPU, or device index out of range */
    RTC_ERR_HIP = -3,       /* a HIP runtime call failed (message has the HIP error string) */
    RTC_ERR_LIMIT = -4,     /* scene exceeds a compiled-in limit (CSG hits, pattern stack, BVH depth) */
    RTC_ERR_UNSUPPORTED = -5,
    RTC_ERR_NOMEM = -6      /* host memory exhausted while building the scene (no C++ exception crosses this ABI) */
} rtc_status;

/* primitive kinds (shapes.py class -> code) */
enum { RTC_SPHERE = 0, RTC_PLANE = 1, RTC_CUBE = 2, RTC_CYLINDER = 3, RTC_CONE = 4,
       RTC_TRIANGLE = 5, RTC_CSG_ROOT = 8, RTC_INSTANCE = 9 };
/* CSG operations (shapes.py:409-422) */
enum { RTC_CSG_UNION = 0, RTC_CSG_INTERSECT = 1, RTC_CSG_DIFFERENCE = 2 };
/* pattern node types (shapes.py:1501-2033) */
enum { RTC_PAT_TEST = 0, RTC_PAT_CHECKERS = 1, RTC_PAT_GRADIENT = 2, RTC_PAT_RADIAL = 3,
       RTC_PAT_RING = 4, RTC_PAT_STRIPE = 5,
       RTC_PAT_PERTURBED = 8, /* only named by RTC_OP_PERTURB; pat_param = frequency, scale factor, octaves */
       RTC_PAT_MAP_PLANAR = 9, RTC_PAT_MAP_SPHERICAL = 10, RTC_PAT_MAP_CYLINDRICAL = 11,
       RTC_PAT_MAP_CUBE = 12, RTC_UV_CHECKER = 20, RTC_UV_ALIGN = 21, RTC_UV_IMAGE = 22 };
/* pattern program opcodes: a material's pattern is a postfix program over a colour stack.
 * EVAL pushes node(point); EVAL_AB pops b then a and pushes node(point) evaluated with those
 * two colours (NestedPattern, shapes.py:1892-1895); BLEND pops two and pushes their mean
 * (BlendedPattern, shapes.py:1874-1877).  PERTURB replaces the world point every later EVAL
 * sees by the point moved with Perlin noise (PerturbedPattern._perturb, shapes.py:1911-1922:
 * node = an RTC_PAT_PERTURBED record), RESTORE brings the previous point back; they bracket the
 * program of the wrapped pattern and nest at most RTC_PERTURB_DEPTH (2) deep.  The noise is
 * pnoise3 of the PyPI `noise` package restated from its published algorithm -- that package is in
 * neither the reference tree nor this image: parity unpinned. */
enum { RTC_OP_EVAL = 0, RTC_OP_EVAL_AB = 1, RTC_OP_BLEND = 2, RTC_OP_PERTURB = 3, RTC_OP_RESTORE = 4 };

/* A "prim code" names something a ray can be tested against:
 *   bits 31..28 kind, bit 27 = does NOT cast shadows, bits 26..0 index
 * index = leaf id for the analytic kinds, triangle-table row for RTC_TRIANGLE,
 * CSG-root row for RTC_CSG_ROOT, row of `inst` for RTC_INSTANCE. */
#define RTC_PRIM(kind, noshadow, index) \
    ((int32_t)(((uint32_t)(kind) << 28) | ((uint32_t)((noshadow) ? 1u : 0u) << 27) | ((uint32_t)(index) & 0x07FFFFFFu)))

/* Flattened scene, as ray_tracer_challenge_amd.compile emits it.  3x4 matrices are the top
 * three rows of the reference's 4x4s, row-major.  "leaf" = one primitive shape object, numbered
 * in depth-first order of world.contains (the id reported in hit-id maps). */
typedef struct {
    /* leaves: Shape objects (shapes.py:228-313, 516-887) */
    int32_t n_leaves;
    const int32_t *leaf_kind;      /* [n_leaves] */
    const int32_t *leaf_xform;     /* [n_leaves] row of `xform` */
    const int32_t *leaf_material;  /* [n_leaves] row of `mat` */
    const int32_t *leaf_tri;       /* [n_leaves] row of the triangle tables or -1 */
    const double *leaf_minmax;     /* [n_leaves][2] Cylinder/Cone minimum, maximum */
    const int32_t *leaf_closed;    /* [n_leaves] */
    /* collapsed world->object transforms: inverse(T_leaf) * ... * inverse(T_root)
     * (Shape.intersect shapes.py:271-273 applied down the parent chain; world_to_object :282-286;
     * its transpose is normal_to_world :288-296) */
    int32_t n_xforms;
    const double *xform;           /* [n_xforms][12] */
    /* triangles, pre-transformed to WORLD space (Triangle/SmoothTriangle shapes.py:799-887) */
    int32_t n_tris;
    const double *tri_geom;        /* [n_tris][12] p1, e1, e2, det_eps = EPSILON*|det(object->world)|, 0, 0 */
    const double *tri_norm;        /* [n_tris][12] world-space (unnormalised) n1, n2, n3, 0,0,0; flat triangles repeat their face normal */
    const int32_t *tri_leaf;       /* [n_tris] leaf id */
    /* things rays are tested against */
    int32_t n_items;               /* bounded: the library builds its AABB BVH over these */
    const int32_t *item_prim;      /* [n_items] prim codes */
    const double *item_lo, *item_hi; /* [n_items][3] world-space AABBs */
    int32_t n_unbounded;           /* planes, uncapped cylinders/cones, CSG with such children: tested for every ray */
    const int32_t *unbounded_prim;
    /* CSG (shapes.py:316-422): every top-most CSG object is a "root"; its nodes are stored
     * children-before-parents; leaves of any subtree are contiguous in leaf order */
    int32_t n_csg_roots;
    const int32_t *csg_root;       /* [n][4] first node, node count, first leaf, end leaf */
    int32_t n_csg_nodes;
    const int32_t *csg_node;       /* [n][5] op, first leaf, first right leaf, end leaf, xform row (world->csg local) */
    const double *csg_bbox;        /* [n][6] CSG.bounds() in csg-local space: min xyz, max xyz */
    /* materials (shapes.py:137-150) */
    int32_t n_materials;
    const double *mat;             /* [n][10] color rgb, ambient, diffuse, specular, shininess, reflective, transparency, refractive_index */
    const int32_t *mat_shadow;     /* [n] casts_shadow */
    const int32_t *mat_program;    /* [n] row of prog_* or -1 for plain colour */
    /* pattern nodes and programs */
    int32_t n_pat_nodes;
    const int32_t *pat_type;       /* [n] */
    const double *pat_inv;         /* [n][12] inverse(pattern.transform) */
    const double *pat_ab;          /* [n][6] colours a, b */
    const int32_t *pat_child;      /* [n][6] uv-pattern nodes of the map types (cube: right,left,up,down,front,back; cylinder: body,cap) */
    const double *pat_param;       /* [n][4] uv checker: width, height */
    const double *pat_colors;      /* [n][15] align check: main, ul, ur, bl, br */
    const int32_t *pat_tex;        /* [n] texture row or -1 */
    int32_t n_programs;
    const int32_t *prog_off;       /* [n] first instruction */
    const int32_t *prog_len;       /* [n] */
    int32_t n_instr;
    const int32_t *instr;          /* [n][2] opcode, pattern node */
    /* image textures (canvas.py:40-55): RGB8, row-major, rows top to bottom */
    int32_t n_textures;
    const int64_t *tex_off;        /* [n] byte offset into tex_data */
    const int32_t *tex_w, *tex_h;  /* [n] */
    const double *tex_scale;       /* [n] the image's largest channel value */
    int64_t tex_bytes;
    const uint8_t *tex_data;
    /* lights (renderer.py:895-973) */
    int32_t n_lights;
    const int32_t *light_type;     /* [n] 0 point, 1 area */
    const double *light_geom;      /* [n][9] position | corner, per-step uvec, per-step vvec */
    const int32_t *light_steps;    /* [n][3] usteps, vsteps, jitter */
    const double *light_intensity; /* [n][3] */
    /* Instanced meshes (the reference loads dragon.obj six times for bounding_boxes.yml:59 ff.: one group of triangles under six
     * transforms).  A mesh is stored ONCE, in the object space its vertices are given in -- where the reference itself runs
     * Moller-Trumbore, shapes.py:812-830, det epsilon = EPSILON as there -- with one tree of its own; an instance is a prim of kind
     * RTC_INSTANCE among the items: the ray is taken to object space by the instance's collapsed transform and walks the mesh's
     * tree.  The leaves of an instance are numbered first_leaf .. first_leaf + count - 1 in row order, their per-leaf entries
     * (leaf_kind = RTC_TRIANGLE, leaf_xform, leaf_material) are filled as for any triangle and leaf_tri = -2 - (mesh row). */
    int32_t n_mesh_tris;
    const double *mesh_geom;       /* [n_mesh_tris][12] object space: p1, e1, e2, EPSILON, 0, 0 */
    const double *mesh_norm;       /* [n_mesh_tris][12] object space n1, n2, n3 (flat triangles repeat their face normal) */
    int32_t n_meshes;
    const int32_t *mesh_range;     /* [n_meshes][2] first row, rows */
    int32_t n_instances;           /* at most RTC_MAX_INSTANCES */
    const int32_t *inst;           /* [n_instances][4] mesh, xform row (world -> the mesh's object space), first leaf, material */
    /* How rtc_scene_create builds the triangle BVH.  0: binned SAH on the host -- the best tree, tens of milliseconds for 10^5
     * triangles: for scenes that are rendered many times.  RTC_BUILD_ON_DEVICE: on the GPU from the uploaded rows (Morton
     * order + parallel locally-ordered clustering, ~2 ms for 1.4 10^5 triangles; frames ~1.1 x the SAH tree's): for scenes
     * rendered once, the reference's own use (main.py:305-354: one frame per scene).  A request, not a different scene: meshes
     * under 16 384 triangles get the host's build anyway (it is quicker than the GPU build's fixed costs), and so does a mesh whose
     * GPU-built tree comes out deeper than the traversal stack (seen at 3.2 10^7 triangles; 8 10^6 fit). */
    int32_t build_flags;
} rtc_scene_desc;
#define RTC_BUILD_ON_DEVICE 1
#define RTC_MAX_INSTANCES 8        /* a ray keeps the instances whose boxes it enters in one 64-bit word, a byte each */

/* Camera (renderer.py:667-692) with its transform already inverted (ray_for_pixel :747-748). */
typedef struct {
    int32_t hsize, vsize;
    double half_width, half_height, pixel_size;
    double inv_transform[12];
} rtc_camera;

/* Run the instrumented kernel variant that also counts work items (slower; for roofline
 * accounting, never for timing). */
#define RTC_FLAG_COUNT_WORK 1
/* The pixel kernel records how long every wave (8x8 pixel block) took and the next frames of
 * the same view on the same scene handle launch the dearest blocks first (any order gives the
 * same image; a good one shortens the tail of the launch).  This flag renders in plain tile
 * order and leaves the recorded costs alone. */
#define RTC_FLAG_NO_FEEDBACK 8

typedef struct {
    int32_t precision;   /* 64 = fp64 (the reference's arithmetic), 32 = fp32: half the registers and bytes, within 1/255 per channel of the fp64 frame
                            on >= 99.9 % of pixels; 1.1 x (mesh walks) to 1.3 x (analytic scenes, patterns) faster, DESIGN.md 4.1 */
    int32_t max_depth;   /* the reference's `remaining`, default 5 (renderer.py:562) */
    int32_t tile_size;   /* pixels per tile edge for rtc_render_tiles*; multiple of 8; 0 -> 32 */
    int32_t flags;       /* RTC_FLAG_* */
    double shade_eps;    /* over/under-point offset (renderer.py:15, :221-222); <= 0 -> 1e-6 in fp64, 1e-5 in fp32 */
    double shade_eps_rel; /* the offset actually used is max(shade_eps, shade_eps_rel * max|coordinate of hit point or ray origin|);
                            < 0 -> default: 0 in fp64 (the reference's fixed EPSILON), 1e-6 (~8 ulp) in fp32 */
    uint64_t seed;       /* jittered area lights only */
} rtc_render_opts;

typedef struct {
    int64_t rays[4];     /* traced: primary, shadow, reflection, refraction */
    int64_t work[8];     /* RTC_FLAG_COUNT_WORK only: BVH node visits (2 box tests each), triangle tests,
                            analytic leaf tests, CSG evaluations, shaded hits, light samples shaded,
                            container passes, pattern evaluations */
    double kernel_ms;    /* trace kernel time measured with HIP events on its stream (host-buffer calls only) */
    int64_t shadow_rays_culled; /* is_shadowed queries answered without a ray: no cell of the light contributes at
                            that hit (light behind the surface), so the result is the ambient term either way */
    int64_t wide_node_visits; /* RTC_FLAG_COUNT_WORK only: how many of work[0]'s node visits were to the triangle BVH's four-wide
                            nodes (4 box tests each; the object BVH's binary nodes: 2) */
} rtc_stats;

typedef struct {
    int32_t device;
    int32_t bvh_nodes, bvh_depth, bvh_max_leaf;
    int64_t device_bytes_fp64, device_bytes_fp32;
} rtc_scene_info;

/* Which trace kernel variant the last frame of a scene handle was launched as (reporting only: bench.py names the kernel its
 * roofline line is about; no counterpart in the reference). */
typedef struct {
    int32_t precision;         /* 64 | 32 */
    int32_t csg;               /* 0 = no CSG / perturbed patterns, 1 = with, 2 = with list-free CSG evaluation */
    int32_t far_camera;        /* rays re-based at the trees' boxes (camera far from the scene) */
    int32_t replicas;          /* the launch order holds split units whose pixels several lanes trace */
    int32_t counting_work;     /* RTC_FLAG_COUNT_WORK variant */
    int32_t waves_per_simd;    /* register budget the variant was built for: 512 / waves registers per lane */
    int32_t workgroup_threads; /* 64: one wave per workgroup */
    int32_t workgroups;        /* grid size */
    int32_t n_units;           /* 8x8 pixel blocks of the frame (or of the tile list) */
    int32_t n_entries;         /* launch-order entries: units, with the split ones as 4 or 16 parts; = n_units in plain order */
    int32_t feedback_stage;    /* frames of this view rendered so far on its workspace (0 = plain order, no feedback) */
} rtc_launch_info;

typedef struct rtc_scene rtc_scene;

int rtc_version(void);
/* number of visible GPUs (0 on a machine without one); never fails */
int rtc_device_count(void);
const char *rtc_last_error(void);
/* Page-locked host memory for result buffers (no counterpart in the reference): rtc_render,
 * rtc_render_tiles and rtc_render_ppm accept any host pointer, but a frame reaches pinned memory
 * at PCIe rate (1080p fp64, 50 MB: 0.9 ms) and pageable memory at a third of it or less. */
int rtc_host_alloc(size_t bytes, void **out);
void rtc_host_free(void *ptr);

/* Validates `desc`, builds the BVH over its items on the host, uploads fp64 + fp32 copies to
 * GPU `device`.  The caller keeps ownership of every array in `desc`. */
int rtc_scene_create(const rtc_scene_desc *desc, int device, rtc_scene **out);
void rtc_scene_destroy(rtc_scene *scene);
/* the triangle BVH rebuilt on the GPU (Morton codes, radix sort, Karras' radix
 * tree, bottom-up boxes) from the triangle rows resident on the device, instead of the host's
 * binned-SAH build of rtc_scene_create.  Same images (any conservative hierarchy gives the
 * reference's hit lists, shapes.py:454); a millisecond-scale build, a tree that costs rays more
 * steps.  RTC_ERR_LIMIT (scene unchanged) if the tree is deeper than the traversal stack.
 * Synchronises the device; do not call while frames of this scene are in flight on other threads. */
int rtc_scene_build_bvh_device(rtc_scene *scene);
/* New positions for every triangle of the scene (the animation loops of main.py:240-265): the
 * tri_geom / tri_norm arrays of rtc_scene_desc again, same triangle count and order, then the GPU
 * rebuild above.  Everything else of the scene stays. */
int rtc_scene_update_triangles(rtc_scene *scene, int64_t n_tris, const double *tri_geom, const double *tri_norm);
/* The same upload, then a REFIT instead of a rebuild: the triangle BVH the scene has now (host SAH
 * or GPU-built) keeps its nodes and leaves, and its boxes are recomputed bottom-up around the moved
 * triangles (one kernel, ~0.1 ms).  Right for deformations and small motions, where the SAH tree's
 * shape stays good; after large relative motion the boxes overlap more and rays slow down (never
 * wrong: the boxes stay conservative), and rtc_scene_update_triangles is the better call.
 * Both keep the recorded launch order of the current view and measure it anew. */
int rtc_scene_refit_triangles(rtc_scene *scene, int64_t n_tris, const double *tri_geom, const double *tri_norm);
/* Either of the two with the new rows already in device memory on the scene's GPU (fp64, same
 * layout; written before this call on the null stream or synchronised by the caller): no PCIe
 * copy.  refit != 0 -> refit, else rebuild. */
int rtc_scene_update_triangles_device(rtc_scene *scene, int64_t n_tris, const double *tri_geom_dev, const double *tri_norm_dev, int refit);
int rtc_scene_get_info(const rtc_scene *scene, rtc_scene_info *out);
/* the variant and grid of the most recent launch on this handle (all zero before the first) */
int rtc_scene_last_launch(const rtc_scene *scene, rtc_launch_info *out);

/* render(cam, world) (renderer.py:753-775).  rgb_out: hsize*vsize*3 doubles in the reference's
 * canvas order [x][y][c] (canvas.py:60); hit_id_out: hsize*vsize leaf ids of the primary hit,
 * -1 = miss, or NULL; stats_out may be NULL.  Host buffers; synchronous. */
int rtc_render(rtc_scene *scene, const rtc_camera *cam, const rtc_render_opts *opts,
               double *rgb_out, int32_t *hit_id_out, rtc_stats *stats_out);

/* render_multi's unit of work (renderer.py:781-793, 811-817): only the listed tiles.  Tiles are
 * tile_size x tile_size pixels, numbered row-major over the image (tile = ty * tiles_x + tx).
 * Output is COMPACT: slot k of rgb_out/hit_id_out holds tile tile_ids[k] as [ty][tx] rows of
 * tile_size pixels (pixels outside the image are left untouched).  Host buffers; synchronous. */
int rtc_render_tiles(rtc_scene *scene, const rtc_camera *cam, const rtc_render_opts *opts,
                     int32_t n_tiles, const int32_t *tile_ids,
                     double *rgb_out, int32_t *hit_id_out, rtc_stats *stats_out);

/* Device-pointer form for callers that keep the frame in HBM (bench.py, the multi-GPU gather).
 * tile_ids_dev == NULL renders the whole frame into the [x][y][c] layout; otherwise the
* compact tile layout above.  rays_dev (may be NULL) is ADDED to and must hold 13 int64 (14 with RTC_FLAG_COUNT_WORK): [0..3] ray
 * counts, [4..11] the work counters of rtc_stats (RTC_FLAG_COUNT_WORK only), [12] shadow rays culled, [13] rtc_stats::wide_node_visits
 * (RTC_FLAG_COUNT_WORK only).  Asynchronous on
 * `stream` (a hipStream_t, NULL = default stream).  A scene handle is immutable while frames are rendered and
 * may be rendered from several host threads at once, one stream per call: every call takes a workspace of its own
 * (the launch-order buffers of RTC_FLAG_NO_FEEDBACK's comment) from the handle and gives it back when its stream
 * has finished with it.  The calls that change a scene (rtc_scene_update_triangles*, rtc_scene_refit_triangles,
 * rtc_scene_build_bvh_device) must not run while frames of it are in flight. */
int rtc_render_device(rtc_scene *scene, const rtc_camera *cam, const rtc_render_opts *opts,
                      int32_t n_tiles, const int32_t *tile_ids_dev,
                      double *rgb_dev, int32_t *hit_id_dev, int64_t *rays_dev, void *stream);

/* rtc_render_device for callers that cannot allocate -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same frame bit for bit; it
 * calls no hipMalloc / hipFree / hipHostMalloc / event creation: the launch-order buffers of RTC_FLAG_NO_FEEDBACK's
 * comment (wave costs, two order lists, per-wave ray count slots: 4.9 MB) live in a device array inside the library, one copy per GPU, held
 * until the process ends.  Limit: 4 194 304 pixels (65 536 blocks of 8x8) per call, RTC_ERR_LIMIT beyond.  Every call
 * on a GPU shares that GPU's array (its costs are cleared by the call that records them; an order list is only read
 * by the view it was sorted for), so calls that overlap in time on one GPU -- different streams, or host threads
 * without stream order between them -- race on it: keep them stream-ordered, or use rtc_render_device.
 * On a stream that is being captured into a HIP graph the call renders in plain tile order and neither reads nor
 * advances the launch-order state (a captured call runs when its graph is replayed, not when it is enqueued). */
int rtc_render_device_noalloc(rtc_scene *scene, const rtc_camera *cam, const rtc_render_opts *opts,
                              int32_t n_tiles, const int32_t *tile_ids_dev,
                              double *rgb_dev, int32_t *hit_id_dev, int64_t *rays_dev, void *stream);

/* construct_ppm_body (canvas.py:133-156) on the device: [x][y][c] fp64 -> [y][x][c] bytes.
 * max_component is 1.0 for default_clamping True, else the canvas maximum (the caller
 * computes it, e.g. with a device-side max over rgb_dev). */
int rtc_quantize_device(const double *rgb_dev, int32_t hsize, int32_t vsize, double max_component,
                        uint8_t *out_dev, void *stream);

/* The canvas from compact tiles (renderer.py:819-829 collects the pool's tiles into the shared
 * canvas, canvas.py:60-61): slot k of tiles_dev ([n_slots][tile_size][tile_size][3] doubles, the
 * layout rtc_render_device writes with a tile list) goes to tile tile_ids_dev[k] of the
 * [x][y][c] frame; slots with a negative tile id (padding of a gathered buffer) and pixels
 * outside the image are skipped.  Device pointers; asynchronous on `stream`. */
int rtc_scatter_tiles_device(const double *tiles_dev, const int32_t *tile_ids_dev, int32_t n_slots, int32_t tile_size,
                             int32_t hsize, int32_t vsize, double *frame_dev, void *stream);

/* The same for the hit-id map: slot k of tiles_dev ([n_slots][tile_size][tile_size] int32, what rtc_render_device writes
 * to hit_id_dev with a tile list) goes to tile tile_ids_dev[k] of the [x][y] id map (no counterpart in the reference,
 * which has no id map; renderer.py:819-829 is the canvas side of it). */
int rtc_scatter_id_tiles_device(const int32_t *tiles_dev, const int32_t *tile_ids_dev, int32_t n_slots, int32_t tile_size,
                                int32_t hsize, int32_t vsize, int32_t *frame_dev, void *stream);

/* max(canvas) of construct_ppm_body's brightness-scaled variant (canvas.py:137-145, np.max over
 * every component): one double to max_out_dev.  n_values = 3 * hsize * vsize.  Asynchronous. */
int rtc_frame_max_device(const double *rgb_dev, int64_t n_values, double *max_out_dev, void *stream);

/* construct_ppm_body (canvas.py:133-156) with max_component read on the device: NULL = 1.0
 * (default_clamping, main.py's final_clamped.ppm), else *max_component_dev (final.ppm, from
 * rtc_frame_max_device).  out_dev: 3 * hsize * vsize bytes, rows of y, then x, then rgb. */
int rtc_ppm_body_device(const double *rgb_dev, int32_t hsize, int32_t vsize, const double *max_component_dev,
                        uint8_t *out_dev, void *stream);

/* render + construct_ppm (main.py:343-354) without the fp64 frame leaving the GPU: traces the
 * frame, encodes it on the device and copies only the P6 files to the host.  clamped_out /
 * scaled_out (either may be NULL) each receive a whole file -- header "P6\n<w> <h>\n255\n",
 * body, trailing newline -- of *length_out bytes; `capacity` is the size of each buffer
 * (rtc_ppm_size() bytes are enough).  Host buffers; synchronous. */
int64_t rtc_ppm_size(int32_t hsize, int32_t vsize);
int rtc_render_ppm(rtc_scene *scene, const rtc_camera *cam, const rtc_render_opts *opts,
                   uint8_t *clamped_out, uint8_t *scaled_out, int64_t capacity, int64_t *length_out,
                   rtc_stats *stats_out);

#ifdef __cplusplus
}
#endif
#endif
