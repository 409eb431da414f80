/* TEST INFRASTRUCTURE ONLY -- not part of the product.
 *
 * CPU oracle for the per-pixel trace loop of gbordelon/ray_tracer_challenge: a literal,
 * brute-force, recursive C restatement of renderer.py / shapes.py (file:line cited at each
 * function in rtc_oracle.c).  It walks the scene TREE with one inverse transform per level,
 * tests every leaf for every ray, keeps full sorted hit lists and recurses for reflection and
 * refraction exactly as the reference does -- i.e. it deliberately shares nothing with the
 * product's flattened/BVH/iterative design, so that product-vs-oracle parity means something.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Parity status: PINNED -- checked against the reference's doctest known-answer vectors
 * (tests/test_oracle_kat.py) and against fixtures rendered by the reference itself in the
 * build container (the .npz files under tests/golden, made by oracle/make_golden.py).
 */
#ifndef RTC_ORACLE_H
#define RTC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SPHERE = 0, ORC_PLANE = 1, ORC_CUBE = 2, ORC_CYLINDER = 3, ORC_CONE = 4,
       ORC_TRIANGLE = 5, ORC_SMOOTH_TRIANGLE = 6, ORC_GROUP = 7, ORC_CSG = 8 };
enum { ORC_UNION = 0, ORC_INTERSECT = 1, ORC_DIFFERENCE = 2 };
enum { ORC_PAT_TEST = 0, ORC_PAT_CHECKERS = 1, ORC_PAT_GRADIENT = 2, ORC_PAT_RADIAL = 3,
       ORC_PAT_RING = 4, ORC_PAT_STRIPE = 5, ORC_PAT_BLENDED = 6, ORC_PAT_NESTED = 7,
       ORC_PAT_PERTURBED = 8, ORC_PAT_MAP_PLANAR = 9, ORC_PAT_MAP_SPHERICAL = 10,
       ORC_PAT_MAP_CYLINDRICAL = 11, ORC_PAT_MAP_CUBE = 12,
       ORC_UV_CHECKER = 20, ORC_UV_ALIGN = 21, ORC_UV_IMAGE = 22 };

typedef struct {
    /* ---- shape tree ---- */
    int32_t n_nodes;
    const int32_t *kind;       /* [n_nodes] */
    const int32_t *parent;     /* [n_nodes] -1 for world.contains entries */
    const double *inv;         /* [n_nodes][16] inverse(node.transform), row-major */
    const int32_t *material;   /* [n_nodes] */
    const double *fparam;      /* [n_nodes][3] minimum, maximum, closed */
    const int32_t *child_off;  /* [n_nodes] group: first child slot; csg: slot of left (right = +1) */
    const int32_t *child_cnt;  /* [n_nodes] */
    const int32_t *child_idx;  /* child slots */
    const int32_t *csg_op;     /* [n_nodes] */
    const double *bbox;        /* [n_nodes][6] node.bounds() min xyz, max xyz (CSG only) */
    const int32_t *tri_idx;    /* [n_nodes] row of `tri` or -1 */
    const double *tri;         /* [n_tri][18] p1 p2 p3 n1 n2 n3 */
    const int32_t *leaf_id;    /* [n_nodes] canonical leaf id, -1 for group/csg */
    int32_t n_top;
    const int32_t *top;        /* world.contains */
    /* ---- materials ---- */
    int32_t n_materials;
    const double *mat;         /* [n][10] color rgb, ambient, diffuse, specular, shininess, reflective, transparency, refractive_index */
    const int32_t *mat_shadow; /* [n] casts_shadow */
    const int32_t *mat_pattern;/* [n] pattern index or -1 */
    /* ---- patterns ---- */
    int32_t n_patterns;
    const int32_t *pat_type;   /* [n] */
    const double *pat_inv;     /* [n][16] inverse(pattern.transform) */
    const double *pat_ab;      /* [n][6]  a rgb, b rgb */
    const int32_t *pat_child;  /* [n][6] */
    const double *pat_param;   /* [n][4] uv checker: width, height; perturbed: freq, scale, octaves */
    const double *pat_colors;  /* [n][15] align check: main ul ur bl br */
    const int32_t *pat_tex;    /* [n] texture index or -1 */
    /* ---- textures ---- */
    int32_t n_textures;
    const int64_t *tex_off;    /* [n] byte offset into tex_data (RGB8 rows) */
    const int32_t *tex_w, *tex_h;
    const double *tex_scale;   /* [n] max over all channel values */
    const uint8_t *tex_data;
    /* ---- lights ---- */
    int32_t n_lights;
    const int32_t *light_type; /* 0 point, 1 area */
    const double *light_geom;  /* [n][9] position|corner, uvec, vvec (per-step vectors) */
    const int32_t *light_steps;/* [n][3] usteps, vsteps, jitter */
    const double *light_intensity; /* [n][3] */
} orc_scene;

typedef struct {
    int32_t hsize, vsize;
    double half_width, half_height, pixel_size;
    double inv_transform[16];  /* inverse(camera.transform) */
} orc_camera;

/* rays[4]: primary, shadow, reflection, refraction -- counted where the reference would trace
 * them.  hit_leaf/hit_t (may be NULL) describe the primary ray's hit: leaf id or -1. */
int orc_render_pixels(const orc_scene *s, const orc_camera *c, int64_t n_pix,
                      const int32_t *px, const int32_t *py, int remaining, int n_threads,
                      double *rgb, int32_t *hit_leaf, double *hit_t, int64_t *rays);

/* known-answer-test entry points (tests/test_oracle_kat.py) */
void orc_ray_for_pixel(const orc_camera *c, int px, int py, double *origin3, double *dir3);
int orc_intersect_world(const orc_scene *s, const double *o3, const double *d3, int cap,
                        double *t_out, int32_t *node_out);
int orc_intersect_node(const orc_scene *s, int node, const double *o3, const double *d3, int cap,
                       double *t_out, int32_t *node_out);
void orc_color_at(const orc_scene *s, const double *o3, const double *d3, int remaining, double *rgb);
void orc_normal_at(const orc_scene *s, int node, const double *p3, double u, double v, double *n3);
/* pnoise3 of the PyPI `noise` package as the reference calls it (shapes.py:1916); restated, parity unpinned */
double orc_pnoise3(double x, double y, double z, int octaves);
void orc_pattern_at_shape(const orc_scene *s, int pattern, int node, const double *p3, double *rgb);
void orc_lighting(const orc_scene *s, int material, int node, int light, const double *p3,
                  const double *eye3, const double *n3, double intensity, double *rgb);
int orc_is_shadowed(const orc_scene *s, const double *light_pos3, const double *p3);
/* comps of the k-th entry of intersect_world(ray): out[0..]: t, point3, eyev3, normalv3, inside,
 * reflectv3, over3, under3, n1, n2, schlick  (23 doubles); returns the hit node or -1 */
int orc_prepare_computations(const orc_scene *s, const double *o3, const double *d3, int k, double *out);

#ifdef __cplusplus
}
#endif
#endif
