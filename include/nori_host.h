/*
 * nori_host.h — C entry points of libnori_host.so, the host side ABOVE the device ABI (include/nori_hip.h):
 * Nori's plugin surface (object factory, XML parser, OBJ loader, SAH BVH build, "path_mis" Integrator plugin)
 * wrapped for callers that are not C++ (tests, bench.py, other-language hosts).  It contains no per-ray code.
 * Functions returning int give 0 on success, -1 on failure with the NoriException text in nori_host_last_error().
 */
#ifndef NORI_HOST_H
#define NORI_HOST_H
#include "nori_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct NoriHostScene NoriHostScene;

const char *nori_host_last_error(void);

/* loadFromXML (src/parser.cpp:28) for a <scene> or a <test type="ttest"> file.  width/height/spp > 0 override the
 * camera size / sampler count of a <scene>.  Tests are loaded without running them.  No GPU is touched. */
NoriHostScene *nori_host_load(const char *xml_path, int width, int height, int spp);
void nori_host_free(NoriHostScene *h);
int nori_host_is_test(NoriHostScene *h);
int nori_host_scene_count(NoriHostScene *h);                 /* 1 for a <scene>, number of child scenes for a <test> */

/* Programmatic scenes (procedural meshes): begin, add meshes, set the camera, end (builds the BVH). */
NoriHostScene *nori_host_scene_begin(void);
int nori_host_scene_add_mesh(NoriHostScene *h, const char *name, const float *V, size_t nV, const uint32_t *F, size_t nF,
                             const float *N, const float *UV, const char *bsdf_type, const float *albedo_or_kd,
                             float alpha, float int_ior, float ext_ior, const float *radiance_or_null);
int nori_host_scene_set_camera(NoriHostScene *h, const float *origin, const float *target, const float *up, float fov, int width, int height);
int nori_host_scene_end(NoriHostScene *h, int spp, int build_threads);

/* info[8]: shapes, primitives, emitters, width, height, spp, reference-layout nodes, 64-byte nodes */
int nori_host_info(NoriHostScene *h, int scene, int *info);
double nori_host_bvh_build_seconds(NoriHostScene *h, int scene);
/* the BVH in the reference's own layout (32-byte nodes: flag|size/axis, start/rightChild, box) + m_indices */
int nori_host_bvh32(NoriHostScene *h, int scene, const void **nodes32, uint32_t *n_nodes, const uint32_t **indices, uint32_t *n_indices);
/* loader output of one shape: counts[4] = nV, nF, hasN, hasUV; pointers stay owned by the scene */
int nori_host_mesh(NoriHostScene *h, int scene, int shape, int *counts, const float **V, const float **N, const float **UV, const uint32_t **F, const float **cdf);
/* flattened tables exactly as handed to nori_hip_upload_scene (valid until the scene is freed) */
const NoriHipSceneDesc *nori_host_flatten(NoriHostScene *h, int scene);

/* Integrator::preprocess: flatten + nori_hip_create(device) + nori_hip_upload_scene.  Needs a GPU. */
int nori_host_preprocess(NoriHostScene *h, int scene, int device);
NoriHipContext *nori_host_context(NoriHostScene *h, int scene);
/* Integrator::render over the whole image, samples [spp_begin, spp_end) */
int nori_host_render(NoriHostScene *h, int scene, int spp_begin, int spp_end);

/* <test type="ttest">: references, sample count, and one scene's test through the GPU integrator.
 * out[7] = mean, variance, reference, t, p-value, alpha, passed */
int nori_host_test_meta(NoriHostScene *h, int scene, float *reference, int *sample_count, float *significance, int *is_path_mis);
int nori_host_test_run(NoriHostScene *h, int scene, double *out);

/* Integrator::Li (include/nori/integrator.h:48-58) of the scene's integrator for n rays (8 floats each: o, mint, d, maxt), one
 * virtual call per ray as a Nori-side caller would make it; 3 floats per ray out.  preprocess must have run. */
int nori_host_li(NoriHostScene *h, int scene, size_t n, const float *rays, float *rgb);

/* The BSDF tests of a <test type="chi2test"> file, or of a <test type="ttest"> file with <bsdf> children (src/chi2test.cpp:93-196,
 * src/ttest.cpp:107-147), run on the device's BSDF code.  Returns the number of tests (or -1); out receives 6 doubles per test, at
 * most max_tests of them: chi2test - chi^2, d.o.f., p-value, alpha, passed, pooled cells; ttest - mean, variance, p-value, alpha,
 * passed, reference.  Needs a GPU. */
int nori_host_bsdf_tests_run(NoriHostScene *h, double *out, int max_tests);

/* h x w x 3 linear rgb floats to a file: "*.exr" = OpenEXR, float R G B, uncompressed scanlines (Bitmap::save, src/bitmap.cpp:82-107),
 * "*.png" = 8-bit sRGB PNG (Bitmap::savePNG, :109-123), anything else = PFM */
int nori_host_write_image(const char *path, const float *rgb, int w, int h);

/* renderScene (src/render.cpp:103): XML in, <scene>.pfm, <scene>.exr and <scene>.png out; out_path receives the .pfm file name */
int nori_host_render_file(const char *xml_path, char *out_path, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif
