/*
 * nori_hip.h — C ABI of libnori_hip.so: the MI355X (gfx950) implementation of Nori's path_mis hot path.
 *
 * This is the drop-in boundary (SURVEY §8 b).  The host side (Nori's plugin surface: XML parser, object
 * factory, OBJ loader, BVH build, Integrator plugin) lives above it and hands over a flattened scene made of
 * plain arrays; everything per-ray happens below it in hand-written HIP kernels.  No C++ types, exceptions or
 * torch types cross this interface.  All functions return 0 on success, a negative NORI_HIP_E* code on failure
 * (nori_hip_last_error() then holds a description for the calling thread).  A context is bound to one GPU and is
 * thread-compatible (one caller at a time); several contexts may coexist, one per device.
 *
 * "Replaces" lines cite the reference interface (paths relative to the reference tree) each entry stands in for.
 */
#ifndef NORI_HIP_H
#define NORI_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NORI_HIP_OK            0
#define NORI_HIP_EINVAL       -1   /* bad argument / inconsistent scene tables */
#define NORI_HIP_EDEVICE      -2   /* HIP runtime error (no device, allocation or launch failure) */
#define NORI_HIP_ESTATE       -3   /* call made in the wrong state (e.g. render before upload) */

#define NORI_HIP_LEAF_BIT   0x80000000u   /* child reference: bit 31 set = leaf, low 31 bits = first record */
#define NORI_HIP_NO_CHILD   0xffffffffu   /* child reference: absent (only in the dummy root of a 1-leaf tree) */
#define NORI_HIP_MISS       0xffffffffu   /* NoriHipHit.prim for a ray that hit nothing */

/* BSDF / shape kinds (src/diffuse.cpp, src/microfacet.cpp, src/dielectric.cpp, src/mirror.cpp, src/subsurface.cpp;
 * src/mesh.cpp, src/sphere.cpp) */
enum { NORI_HIP_BSDF_DIFFUSE = 0, NORI_HIP_BSDF_MICROFACET = 1, NORI_HIP_BSDF_DIELECTRIC = 2, NORI_HIP_BSDF_MIRROR = 3,
       NORI_HIP_BSDF_SUBSURFACE = 4 };
enum { NORI_HIP_SHAPE_MESH = 0, NORI_HIP_SHAPE_SPHERE = 1 };
/* integrators (src/pathMisIntegrator.cpp — the hot path; src/pathMatsIntegrator.cpp — the other path integrator of the
 * reference's own test files, BSDF sampling only; src/directMisIntegrator.cpp — one bounce, emitter + BSDF sampling) */
enum { NORI_HIP_INTEGRATOR_PATH_MIS = 0, NORI_HIP_INTEGRATOR_PATH_MATS = 1, NORI_HIP_INTEGRATOR_DIRECT_MIS = 2 };
/* image kinds and lookup modes (cuda/nori/imagetexture.cpp.cu, bumpMap.cpp.cu, normalMap.cpp.cu; src/imagetexture.cpp) */
enum { NORI_HIP_TEX_COLOR = 0, NORI_HIP_TEX_BUMP = 1, NORI_HIP_TEX_NORMAL = 2,
       /* procedural and scalar textures of the CPU renderer (evaluated on the device from a few constants, no image): */
       NORI_HIP_TEX_CHECKER_COLOR = 3,   /* Checkerboard<Color3f>, "checkerboard_color" (src/checkerboard.cpp:32-55): an albedo */
       NORI_HIP_TEX_CHECKER_FLOAT = 4,   /* Checkerboard<float>, "checkerboard_float"; ConstantTexture<float>, "constant_float" (value1 = value2) */
       NORI_HIP_TEX_IMAGE_FLOAT = 5 };   /* ImageTexture<float>, "image_float" (src/imagetexture.cpp:26-46): grey image, truncating lookup, v / 255 */
enum { NORI_HIP_LOOKUP_NEAREST = 0, NORI_HIP_LOOKUP_BILINEAR = 1 };

/*
 * 64-byte BVH node: one inner node of the reference tree with BOTH children's boxes, so one aligned 64-B fetch
 * decides both descents.  Replaces BVH::BVHNode (include/nori/bvh.h:127-164, 32 B, one box per node).  Node i's
 * children are the reference's left (idx+1) and right (rightChild) nodes; boxes are bit-identical to theirs.
 * Inner nodes are numbered in the reference's pre-order, node 0 is the root.
 */
typedef struct NoriHipNode {
    float    lmin[3], lmax[3];     /* left child's box  */
    float    rmin[3], rmax[3];     /* right child's box */
    uint32_t left, right;          /* child references (inner index | NORI_HIP_LEAF_BIT+first record | NORI_HIP_NO_CHILD) */
    uint32_t lcount, rcount;       /* leaf sizes (0 for inner children); informational, kernels use the record flags */
} NoriHipNode;

/*
 * 48-byte primitive record, stored in the order of the reference's m_indices array (BVH leaf order), so the
 * primitives of a leaf are contiguous.  Replaces the m_indices -> findShape -> m_F -> m_V gather chain of
 * BVH::rayIntersect / Mesh::rayIntersect (src/bvh.cpp:435-449, src/mesh.cpp:84-89).
 *   triangle: a = p0, e1 = p1 - p0, e2 = p2 - p0 (fp32 differences, as the reference computes per test)
 *   sphere  : a = centre, e1[0] = radius
 */
typedef struct NoriHipPrim {
    float    a[3];
    float    e1[3];
    float    e2[3];
    uint32_t prim;                 /* global primitive id = the value in the reference's m_indices */
    uint32_t shape_flags;          /* bits 0..23 shape index, bit 30 = sphere, bit 31 = last record of its leaf */
    uint32_t pad;
} NoriHipPrim;

/* 96-byte per-primitive shading data, same order as NoriHipPrim.  Replaces Mesh::setHitInformation's gathers
 * (src/mesh.cpp:123-168).  Normals / uvs are ignored unless the owning shape has them. */
typedef struct NoriHipShadePrim {
    float p0[3], p1[3], p2[3];
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
} NoriHipShadePrim;

/* 64 bytes: one cache line per shape */
typedef struct NoriHipShape {
    int32_t  type;                 /* NORI_HIP_SHAPE_* */
    int32_t  bsdf;                 /* NORI_HIP_BSDF_* */
    int32_t  emitter;              /* index into emitters, -1 if none */
    uint32_t flags;                /* bit0 has vertex normals, bit1 has uvs */
    float    albedo[3];            /* diffuse / subsurface albedo | microfacet kd (constant colour) */
    union { float alpha;           /* microfacet */
            float ss_cs; };        /* subsurface: colorNorm(s) (src/subsurface.cpp:91-97) */
    float    int_ior, ext_ior;
    union { float ks;              /* microfacet: 1 - max(kd) */
            float ss_cd; };        /* subsurface: colorNorm(dmfp) (src/subsurface.cpp:108-119) */
    float    center[3], radius;    /* sphere */
    uint16_t albedo_tex;           /* 0: `albedo` is the colour; k: textures[k-1] (NORI_HIP_TEX_COLOR) gives it at the hit's uv */
    uint16_t normal_tex;           /* 0: none; k: textures[k-1] (NORI_HIP_TEX_BUMP / _NORMAL) replaces the shading normal */
} NoriHipShape;

/*
 * One image, 4 bytes per texel, rows x cols, row-major at tex_data + offset.  Replaces the cv::Mat members and
 * cudaTextureObject_t of ImageTexture<Color3f>, BumpMap and NormalMap (cuda/nori/imagetexture.cpp.cu:67-104,
 * bumpMap.cpp.cu:55-115, normalMap.cpp.cu:52-97).
 *   NORI_HIP_TEX_COLOR / _NORMAL : texel = R, G, B, A bytes (the fork converts BGR -> RGBA before upload)
 *   NORI_HIP_TEX_BUMP            : texel = two int16: Sobel d/dx, d/dy of the grey image (cv::Sobel, CV_16S, ksize 3,
 *                                  scale = height, delta = grad_delta), read as value / 32767
 *   NORI_HIP_TEX_CHECKER_COLOR   : no image: width = 6, height = 1, the "texels" are six floats, value1 rgb then value2 rgb; the
 *                                  colour at uv is Checkerboard::eval's (src/checkerboard.cpp:32-55: fmod of uv / scale - delta by 2)
 *   NORI_HIP_TEX_CHECKER_FLOAT   : the same with width = 2: value1, value2 (a constant_float has value1 = value2)
 *   NORI_HIP_TEX_IMAGE_FLOAT     : texel = R, G, B, A bytes with R = the grey value; row = x * rows, col = y * cols, value / 255
 * The two float kinds can only be a shape's normal_tex: the CPU renderer's bump_map (src/bumpMap.cpp:44-50) takes a
 * Texture<float> child and, as written, ASSIGNS its value v to the shading normal - n = (v, v, v), not normalised - on meshes
 * that have vertex normals (src/mesh.cpp:151-164; Sphere::setHitInformation does not look at the modifier).  Restated as written.
 * lookup: NORI_HIP_LOOKUP_BILINEAR is the CUDA fork's tex2D (wrap addressing, normalised coordinates, 8-bit blend
 * weights) and, for colour images, its sRGB -> linear conversion; NORI_HIP_LOOKUP_NEAREST is the CPU renderer's
 * lookup for colour images (src/imagetexture.cpp:49-77: row = x*rows, col = y*cols, no colour-space conversion).
 */
typedef struct NoriHipTexture {
    uint32_t kind, lookup;
    uint32_t width, height;        /* image.cols, image.rows */
    uint64_t offset;               /* byte offset of texel (row 0, col 0) in tex_data; multiple of 4 */
    float    scale[2], delta[2];   /* uv -> (uv / scale) - delta, then the fractional part (imagetexture.cpp.cu:15-20) */
} NoriHipTexture;

/* Area emitter bound to a shape (src/arealight.cpp).  Mesh emitters sample a triangle through the normalised
 * area CDF (include/nori/dpdf.h) stored at cdf[cdf_offset .. cdf_offset + n_tris], and read its vertices from
 * light_prims[light_prim_offset + triangle] (mesh order, not BVH order). */
typedef struct NoriHipEmitter {
    float    radiance[3];
    int32_t  shape;
    uint32_t cdf_offset, n_tris, light_prim_offset;
    float    area_pdf;             /* DiscretePDF::getNormalization() = 1 / total area */
} NoriHipEmitter;

typedef struct NoriHipCamera {
    float    sample_to_camera[16]; /* row-major 4x4 (src/perspective.cpp:78-80) */
    float    camera_to_world[16];  /* row-major 4x4 */
    int32_t  width, height;
    float    near_clip, far_clip;
    float    filter_radius;        /* reconstruction filter radius: film border = ceil(radius - 0.5) (src/block.cpp:57) */
} NoriHipCamera;

/*
 * Environment emitter: a lat-long image around the scene (the CUDA fork's <enviromentMap type="sphereMap">,
 * cuda/nori/spheremap.cpp.cu, cuda/include/nori/scene.h:79; the CPU renderer has none).  A ray that leaves the scene in world
 * direction d sees textures[texture - 1] at uv = (acos(p.z) / pi, (atan2(p.y, p.x) + pi) / (2 pi)), p = d in the frame below, times
 * `brightness` (SphereMap::eval, :61-70), with the MIS weight of the last BSDF sample as for a hit emitter
 * (cuda/nori/pathMisIntegrator.cpp.cu:46-75); emitter sampling picks the map with probability 1 / (emitters + 1) (:176-184).
 * The fork's own sample() / pdf() are unfinished (pdf is marked "todo", sample returns radiance that was never divided by a
 * density and builds directions that do not invert eval's mapping): this library samples the map's (u, v) cells in proportion to
 * (r + g + b) * sin(theta) from a table it builds at upload, with the matching solid-angle density.  Parity unpinned.
 */
typedef struct NoriHipEnvMap {
    uint32_t texture;              /* 0: no environment emitter; k: textures[k-1] (NORI_HIP_TEX_COLOR) */
    float    brightness;           /* brightness_factor */
    float    frame[9];             /* SphereMap::origin = Frame(origin vector): s, t, n (cuda/nori/spheremap.cpp.cu:33-34) */
} NoriHipEnvMap;

typedef struct NoriHipSceneDesc {
    uint32_t                n_nodes;   const NoriHipNode      *nodes;
    uint32_t                n_prims;   const NoriHipPrim      *prims;
                                       const NoriHipShadePrim *shade_prims;   /* n_prims */
    uint32_t                root;      /* child reference of the root (inner 0, or a leaf when the tree is one leaf) */
    float                   root_min[3], root_max[3];                          /* root box (src/bvh.cpp:342) */
    uint32_t                n_shapes;  const NoriHipShape     *shapes;
    uint32_t                n_emitters; const NoriHipEmitter  *emitters;
    uint32_t                n_cdf;     const float            *cdf;
    uint32_t                n_light_prims; const NoriHipShadePrim *light_prims;
    NoriHipCamera           camera;
    uint32_t                n_textures; const NoriHipTexture  *textures;      /* may be 0 / NULL */
    uint64_t                n_tex_bytes; const uint8_t        *tex_data;
    NoriHipEnvMap           env;                                               /* env.texture = 0: none */
} NoriHipSceneDesc;

/* ray in, 32 B: replaces Ray3f (include/nori/ray.h:38-100); dRcp is recomputed on the device */
typedef struct NoriHipRay { float o[3], mint, d[3], maxt; } NoriHipRay;
/* hit out, 16 B: global primitive id (NORI_HIP_MISS if none), t, and the uv that Intersection::uv holds after
 * setHitInformation (barycentrics, or interpolated texture coordinates when the mesh has them) */
typedef struct NoriHipHit { uint32_t prim; float t, u, v; } NoriHipHit;

typedef struct NoriHipStats {
    uint64_t camera_samples;       /* samples started */
    uint64_t closest_rays;         /* extend stage; includes the probe rays of subsurface hits (src/pathMisIntegrator.cpp:88-95) */
    uint64_t shadow_rays;          /* shadow stage */
    uint64_t invalid_samples;      /* dropped by the Color3f::isValid rule (src/block.cpp:94-98) */
    uint64_t boxes_tested;         /* slab tests performed (two per 64-byte node fetched, plus the root box) */
    uint64_t prims_tested;         /* ray/primitive tests performed */
    uint64_t iterations;           /* wavefront iterations of the last render call */
    double   render_ms;            /* device time of the last render call (HIP events on the context stream) */
    double   extend_ms;            /* device time of the extend-stage launches (k_trace_pool<closest>) since the last reset_stats, HIP events */
    uint64_t extend_launches;      /* how many launches extend_ms covers (0: the single-kernel pipeline ran) */
    uint64_t extend_rays, extend_boxes, extend_prims;   /* work of those launches */
    uint64_t extend_visits;        /* node fetches of those launches (four- and eight-box trace stages; 0 for the two-box walk, which fetches extend_boxes / 2) */
    uint64_t extend_width;         /* boxes per node the trace stage of the last render walked: 2, 4, or 8 (quantised boxes) */
    uint64_t shadow_rays_culled;   /* emitter samples whose term was exactly zero and whose shadow ray was therefore not traced (NORI_HIP_CULL=0: none); the
                                      reference traces them (src/pathMisIntegrator.cpp:125): shadow_rays + shadow_rays_culled is its shadow-ray count */
    uint64_t ss_paths_dropped;     /* subsurface bounces whose 4096 probe rays all left the shape: the reference would go on probing
                                      (src/pathMisIntegrator.cpp:88-90), here the path ends and the radiance it had gathered so far IS splatted
                                      with full weight (a bias, counted here so that it is never silent; 0 on every scene the reference ships) */
    uint64_t extend_node_bytes;    /* size of the node the trace stage of the last render fetched per visit: 64 (two boxes, or four quantised
                                      boxes: device-built trees with placed leaves, NORI_HIP_QUANT) or 128 (four float boxes, eight quantised) */
    uint64_t pipelines;            /* how the last render call ran its wavefront stages: 1, or 2 = as two pipelines side by side, each over half of the
                                      call's sample range with a pool and stream of its own (trees beyond the caches, NORI_HIP_PIPELINES); every counter
                                      above covers both, and extend_ms / extend_launches are then launches that shared the GPU with the other pipeline's */
} NoriHipStats;

typedef struct NoriHipContext NoriHipContext;

const char *nori_hip_last_error(void);

/* One context per GPU.  Replaces the implicit process-wide state of the reference's renderer
 * (cudaSetDevice(0) in cuda/CudaRender.cpp:17; Scene ownership in src/render.cpp:112-123). */
int nori_hip_device_count(int *n);
int nori_hip_create(int device, NoriHipContext **out);
int nori_hip_destroy(NoriHipContext *ctx);

/* Copies the flattened scene to the device and (re)allocates the film.  Replaces the per-object cudaMalloc /
 * cudaMemcpy marshalling of cuda/CudaRender.cpp:80-98 and ImageBlock::init (src/block.cpp:38-74). */
int nori_hip_upload_scene(NoriHipContext *ctx, const NoriHipSceneDesc *desc);

/* Same, but the library builds the tree itself, on the device: a linear BVH (Morton order of the records' box centres,
 * Karras 2012) with at most NORI_HIP_LBVH_LEAF (default 3) records per leaf.  desc->prims / shade_prims may come in any
 * order (bit 31 of shape_flags is ignored); nodes, n_nodes, root, root_min and root_max are ignored.  No counterpart
 * in the reference, whose SAH build (src/bvh.cpp:54-306) runs on the host and is what nori_hip_upload_scene expects to
 * have been done: this entry is SURVEY §8f's "GPU BVH build" for scenes whose host build would dominate (10 M
 * triangles: 14 s).  The tree is not the reference's: closest hits name the same primitive except where two
 * primitives tie in t, any-hit answers are identical; traversal enters the nearer child first.
 * nori_hip_build_info: size of the node array, nodes in use, tree depth, device time of the build. */
int nori_hip_upload_scene_build(NoriHipContext *ctx, const NoriHipSceneDesc *desc);
int nori_hip_build_info(NoriHipContext *ctx, uint32_t *n_nodes, uint32_t *n_used, uint32_t *depth, double *build_ms);

/* Which Li the context's nori_hip_li_samples / nori_hip_render evaluate (default NORI_HIP_INTEGRATOR_PATH_MIS).  Replaces
 * the choice of Integrator subclass by the scene file's <integrator type="..."> (NoriObjectFactory, src/parser.cpp).
 * path_mats and direct_mis render through the single-kernel pipeline. */
int nori_hip_set_integrator(NoriHipContext *ctx, int kind);

/* Film accumulation mode.  table33 = the tabulated reconstruction filter ImageBlock::init builds (src/block.cpp:54-64:
 * filter->eval(radius * i / 32) for i < 32, then 0): samples are then splatted onto every pixel within `radius` with weight
 * w(dx) * w(dy), which is ImageBlock::put as upstream Nori has it (src/block.cpp:113-128).  In this fork that code is dead
 * behind an early return (:107-111), and the single-pixel add before it is the default here too: table33 = NULL restores it.
 * radius must not exceed the uploaded camera's filter_radius (the film border is sized from that). */
int nori_hip_set_film_filter(NoriHipContext *ctx, const float *table33, float radius);

/* Closest-hit (any_hit = 0) or any-hit (any_hit != 0: hits[i].prim is 1 when occluded, NORI_HIP_MISS otherwise)
 * queries for n host rays.  Replaces Scene::rayIntersect -> BVH::rayIntersect (src/bvh.cpp:404-463) including
 * Shape::setHitInformation's uv (src/mesh.cpp:137-141).  `full`, if not NULL, receives 16 floats per ray:
 * p.xyz, uv.xy, shading normal xyz, geometric normal xyz, shape index, 4 x 0 (Intersection, include/nori/shape.h:38-67).
 * The rays walk the node layout the render's trace stage walks for this scene where that layout changes which boxes a ray enters
 * (quantised four- or eight-box nodes of device-built trees), so that hit parity is tested on the traversal that renders. */
int nori_hip_trace(NoriHipContext *ctx, const NoriHipRay *rays, size_t n, int any_hit, NoriHipHit *hits, float *full);

/* Same query on rays already resident in device memory (device pointers); asynchronous on the context's stream. */
int nori_hip_trace_device(NoriHipContext *ctx, const NoriHipRay *d_rays, size_t n, int any_hit, NoriHipHit *d_hits);

/* Primary rays for n pixel-sample positions.  Replaces PerspectiveCamera::sampleRay (src/perspective.cpp:90-111). */
int nori_hip_camera_rays(NoriHipContext *ctx, const float *pixel_samples_xy, size_t n, NoriHipRay *rays);

/* Radiance of n individual camera samples: sample i seeds pcg32 with (states[i], seqs[i]), draws the pixel jitter
 * and the aperture sample for pixel (px[i], py[i]) as renderBlock does (src/render.cpp:85-86) — or, when `rays`
 * is not NULL, uses rays[i] instead of the camera after consuming the same four draws; or, when px, py and rays
 * are all NULL, places the sample anywhere on the image, pixelSample = next2D() * (width, height), as the t-test
 * harness does (src/ttest.cpp:171-173) — and runs PathMisIntegrator::Li (src/pathMisIntegrator.cpp:18-168).
 * Writes 3 floats per sample. */
int nori_hip_li_samples(NoriHipContext *ctx, size_t n, const int32_t *px, const int32_t *py, const uint64_t *states,
                        const uint64_t *seqs, const NoriHipRay *rays, float *rgb);

/* The device's BSDF code alone, n queries of one constant-colour BSDF (bsdf->bsdf and its parameters; the shape fields of the
 * record are ignored).  wo == NULL: BSDF::sample(bRec(wi[i]), sample[i], sample2[i]) (sample2 NULL: sample again; only the
 * subsurface model reads it), then pdf and eval at the sampled direction.  wo != NULL: pdf and eval at (wi[i], wo[i]).
 * out: 12 floats per query - weight rgb, wo xyz, pdf, eval rgb, isDelta, wo_valid (0: the reference's sample() returned early
 * and left wo unset).  Replaces the direct BSDF calls of the reference's test plugins: the BSDF mode of the t-test
 * (src/ttest.cpp:107-147) and the chi^2 test of sample() against pdf() (src/chi2test.cpp:93-196); the host plugins "ttest"
 * and "chi2test" run on it.  No scene needs to be uploaded. */
int nori_hip_bsdf_vectors(NoriHipContext *ctx, const NoriHipShape *bsdf, size_t n, const float *wi, const float *sample, const float *sample2,
                          const float *wo, float *out);
/* The device's texture code alone: textures[texture] of the uploaded scene (0-based) at n uv pairs.  Colour kinds (NORI_HIP_TEX_COLOR,
 * _CHECKER_COLOR): out = the colour, Texture<Color3f>::eval(uv) (cuda/nori/imagetexture.cpp.cu:12-26, src/checkerboard.cpp:32-55).
 * Normal-modifier kinds: out = the new shading normal for the shading normal nrm[i] (3 floats each, required), NormalModifier::eval /
 * adjustNormal (cuda/nori/bumpMap.cpp.cu:10-27, normalMap.cpp.cu:10-24, src/bumpMap.cpp:44-50).  3 floats per query.  Test entry:
 * the reference has no counterpart call, its textures are only reached through BSDF::eval and setHitInformation. */
int nori_hip_texture_eval(NoriHipContext *ctx, uint32_t texture, size_t n, const float *uv, const float *nrm, float *out);
/* n draws of the device's pcg32 seeded (state, seq): raw 32-bit words (out_u32) or nextFloat() (out_f32), exactly one of the two.
 * Known-answer test entry for ext/pcg32/pcg32-demo.out; Independent::next1D (src/independent.cpp:55-60). */
int nori_hip_pcg32(NoriHipContext *ctx, uint64_t state, uint64_t seq, size_t n, uint32_t *out_u32, float *out_f32);

/* Renders samples [spp_begin, spp_end) of every pixel of the tile [x0,x1) x [y0,y1) and ACCUMULATES them into
 * the film.  Sample s of pixel (x, y) uses the stream pcg32.seed(initstate = y*width + x, initseq = s).
 * Replaces the body of RenderThread::renderScene's pass loop: renderBlock + ImageBlock::put
 * (src/render.cpp:151-190, src/block.cpp:93-111).
 * Over a tree that does not fit the caches (or with NORI_HIP_PIPELINES=2) the call runs as two pipelines side by side: the second half
 * of the sample range (of the pixels, when the range is one sample) goes through a second pool on a second stream, queued by a helper thread that lives for the duration of the call;
 * the context's stream waits for that stream, so whatever the caller queues behind the call (a film read, a filter pass) still sees the
 * whole result.  Same samples, same film; NoriHipStats.pipelines says what ran.  The second pool is allocated on first use and kept.
 * A call that returns an error may have added a part of its samples to the film (as a call that fails half-way always could): clear or
 * discard the film before rendering on. */
int nori_hip_render(NoriHipContext *ctx, int x0, int y0, int x1, int y1, int spp_begin, int spp_end);

/* The same for a list of n_tiles tiles (x0, y0, x1, y1 each; at most 65536, not overlapping) in ONE pass: one pool fill, one
 * tail, whatever the number of tiles.  This is what a rank of a tile-partitioned job calls with its share of the image blocks.
 * Replaces the reference's loop over the blocks a worker thread is handed by the BlockGenerator
 * (src/render.cpp:157-183, src/block.cpp:163-195). */
int nori_hip_render_tiles(NoriHipContext *ctx, const int *tiles_xyxy, int n_tiles, int spp_begin, int spp_end);

int nori_hip_clear_film(NoriHipContext *ctx);
/* film = (height + 2*border) x (width + 2*border) float4 (r, g, b, weight) sums, row-major: ImageBlock's own
 * layout (include/nori/block.h:48).  nori_hip_film_dims reports rows, cols, border. */
int nori_hip_film_dims(NoriHipContext *ctx, int *rows, int *cols, int *border);
int nori_hip_read_film(NoriHipContext *ctx, float *film_rgba);
int nori_hip_write_film(NoriHipContext *ctx, const float *film_rgba);
/* device pointer of the film, for in-place collectives (RCCL reduce over xGMI) by the caller */
int nori_hip_film_device_ptr(NoriHipContext *ctx, void **d_film, size_t *bytes);
/* height x width x 3 floats, rgb / weight.  Replaces ImageBlock::toBitmap (src/block.cpp:76-82). */
int nori_hip_read_bitmap(NoriHipContext *ctx, float *rgb);

/* NL-means filter pass over the current bitmap, on sRGB-encoded colours as the CUDA fork does
 * (cuda/filter.cc.cu:53-110, window r, patch f, sigma, h; scene parameters cuda/nori/scene.cpp:28-32).
 * Output: height x width x 3 floats, sRGB-encoded (what the fork stores in `image`). */
int nori_hip_filter(NoriHipContext *ctx, int r, int f, float sigma, float h, float *rgb_srgb_out);

/* nori_hip_filter for callers that cannot allocate (code under HIP graph capture, tight progressive-display loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  The same kernels and the same result as nori_hip_filter (its
 * default, separable form), but: the output goes to a DEVICE buffer of the caller (height x width x 3 floats), the call
 * neither allocates nor frees nor synchronises (it is asynchronous on the context's stream), and the intermediate sRGB
 * bitmap lives in a static device array of NORI_HIP_NOALLOC_MAX_PIXELS x 3 floats (24.9 MB, one copy per GPU, held until
 * the process ends).  Images with more pixels are rejected with NORI_HIP_EINVAL.  All calls on one GPU share that array:
 * calls that overlap on one GPU (two contexts, two threads) race. */
#define NORI_HIP_NOALLOC_MAX_PIXELS 2073600   /* 1920 x 1080 */
int nori_hip_filter_noalloc(NoriHipContext *ctx, int r, int f, float sigma, float h, float *d_rgb_srgb_out);

int nori_hip_stats(NoriHipContext *ctx, NoriHipStats *out);
int nori_hip_reset_stats(NoriHipContext *ctx);
/* stream the context launches on (hipStream_t as void*), for callers that time or order work around it */
int nori_hip_stream(NoriHipContext *ctx, void **stream);
int nori_hip_synchronize(NoriHipContext *ctx);

/*
 * Several GPUs of one node behind one handle.  No counterpart in the reference, which renders on one device
 * (cudaSetDevice(0), cuda/CudaRender.cpp:17); this is how a Nori-side caller (the "path_mis" plugin's `devices` property, the
 * headless CLI) uses the node.  A group owns one context per listed device.  The scene is uploaded to every device;
 * nori_hip_group_render gives every device its share of samples [spp_begin, spp_end) -
 *   NORI_HIP_PARTITION_SPP    a contiguous slice of the sample range for the whole image, or
 *   NORI_HIP_PARTITION_TILES  every n-th band of 32 rows (NORI_BLOCK_SIZE, include/nori/block.h:30), all samples, one pass -
 * renders them concurrently (one host thread per device) and sums the fp32 films in place with one RCCL all-reduce over xGMI:
 * ImageBlock::put(ImageBlock&) (src/block.cpp:131-140) across devices.  Afterwards every member's film is the job's film:
 * read it, or run nori_hip_filter, on nori_hip_group_context(g, 0).  Sample streams are keyed by the global sample index, so
 * the result does not depend on the number of devices beyond the order of the fp32 sums.  Like nori_hip_render the call
 * ACCUMULATES: what the films held before it (the sum of earlier group renders) is counted once.  RCCL is loaded when the first
 * group is created.  A group is driven by one caller at a time.
 */
typedef struct NoriHipGroup NoriHipGroup;
enum { NORI_HIP_PARTITION_SPP = 0, NORI_HIP_PARTITION_TILES = 1 };
int nori_hip_group_create(const int *devices, int n, NoriHipGroup **out);
int nori_hip_group_destroy(NoriHipGroup *g);
int nori_hip_group_size(NoriHipGroup *g);
NoriHipContext *nori_hip_group_context(NoriHipGroup *g, int i);
int nori_hip_group_upload_scene(NoriHipGroup *g, const NoriHipSceneDesc *desc, int build_on_device);
int nori_hip_group_set_integrator(NoriHipGroup *g, int kind);
int nori_hip_group_set_film_filter(NoriHipGroup *g, const float *table33, float radius);
int nori_hip_group_clear_film(NoriHipGroup *g);
int nori_hip_group_render(NoriHipGroup *g, int spp_begin, int spp_end, int partition);
/* The work split of nori_hip_group_render as a pure function (no device is touched; usable for planning and tested on the CPU):
 * what member `member` of `n_members` renders of samples [spp_begin, spp_end) of an image `height` rows high.
 *   NORI_HIP_PARTITION_SPP    *out_spp_begin / *out_spp_end = its contiguous slice (the first (count mod n) members one sample longer;
 *                             empty, begin == end, for members beyond the sample count), *out_n_bands = 0
 *   NORI_HIP_PARTITION_TILES  the whole sample range, and its bands of 32 rows (band k belongs to member k mod n): *out_n_bands of
 *                             them, the first min(*out_n_bands, max_bands) written to bands_y0y1 as (y0, y1) pairs (bands_y0y1 may be NULL)
 * Over the members the slices tile the sample range, and the bands tile the rows, exactly once. */
int nori_hip_group_plan(int n_members, int member, int spp_begin, int spp_end, int height, int partition,
                        int *out_spp_begin, int *out_spp_end, int *bands_y0y1, int max_bands, int *out_n_bands);
/* If a member fails inside nori_hip_group_render, the films of ALL members are cleared before the error is returned (a member may
 * have rendered part of its share and the others have dropped what they held): the caller renders the job again from sample 0. */
/* the all-reduce alone (for callers that render on the member contexts themselves), and its device time on the last call */
int nori_hip_group_reduce_films(NoriHipGroup *g);
int nori_hip_group_reduce_ms(NoriHipGroup *g, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* NORI_HIP_H */
