// Scene graph classes behind the plugin names the reference registers for the path_mis hot path.
//
// Reference: include/nori/{shape,mesh,bsdf,emitter,camera,sampler,integrator,scene,bvh}.h.
// A BSDF / Emitter / Shape here is a parameter holder: its per-ray methods (sample/eval/pdf, rayIntersect, ...)
// run on the GPU, reached through Integrator::render() -> include/nori_hip.h.  describe() is the one thing each
// plugin must provide: how it flattens into the C ABI's tables.
#pragma once
#include "nori.h"

namespace nori {

/// An image as the device reads it (include/nori_hip.h: NoriHipTexture + 4 bytes per texel); `meta.offset` is set
/// when the scene is flattened.
struct ImageData { NoriHipTexture meta; std::vector<uint8_t> texels; };
/// Decodes a binary PNM (P5 grey / P6 colour, 8 bit), `filename` being looked up through the file resolver as the
/// reference does (cuda/nori/imagetexture.cpp.cu:36).  For any other format the pre-decoded sibling "<file>.ppm" /
/// "<file>.pgm" is read instead (the reference decodes with cv::imread, which this build does not have).
/// Returns rows x cols x channels bytes, channels = 1 or 3 (RGB order).
void loadImageFile(const std::string &filename, int &cols, int &rows, int &channels, std::vector<uint8_t> &data);
/// Reads a Wavefront OBJ file into the arrays Mesh holds (host/objreader.cpp): V 3 per vertex, N / UV if the file has them, F 3 per
/// triangle; vertex numbering, quad split, toWorld and de-duplication as src/obj.cpp:32-132 produces them.
void readWavefrontOBJ(const std::string &filename, const Transform &trafo, std::vector<float> &V, std::vector<float> &N, std::vector<float> &UV,
                      std::vector<uint32_t> &F, BoundingBox3f &bbox);

/// Texture<Color3f> (include/nori/texture.h): a constant colour or an image
class ColorTexture : public NoriObject {
public:
    EClassType getClassType() const override { return ETexture; }
    virtual Color3f constantValue() const { return Color3f(0.f); }
    virtual const ImageData *image() const { return nullptr; }
    /// true for the Texture<float> plugins (constant_float, checkerboard_float, image_float): they share the <texture> tag and class
    /// type with the colour textures, and a slot that wants the other kind says so instead of reinterpreting the object
    virtual bool scalar() const { return false; }
};
/// NormalModifier (include/nori/normalModifier.h; cuda/include/nori/bumpMap.h, normalMap.h)
class NormalModifier : public NoriObject {
public:
    EClassType getClassType() const override { return ENormalModifier; }
    virtual const ImageData *image() const = 0;
};

class BSDF : public NoriObject {
public:
    EClassType getClassType() const override { return EBSDF; }
    virtual bool isDelta() const = 0;
    virtual void describe(NoriHipShape &out) const = 0;      // fills bsdf kind + parameters
    virtual const ImageData *albedoImage() const { return nullptr; }
};

class Shape;
class Emitter : public NoriObject {
public:
    EClassType getClassType() const override { return EEmitter; }
    void setShape(Shape *s) { m_shape = s; }
    const Shape *getShape() const { return m_shape; }
    virtual Color3f getRadiance() const = 0;
protected:
    Shape *m_shape = nullptr;
};

class Shape : public NoriObject {
public:
    ~Shape() override;
    EClassType getClassType() const override { return EMesh; }
    void addChild(NoriObject *child) override;               // src/shape.cpp:41-69
    void activate() override;                                // default diffuse BSDF, src/shape.cpp:32-39
    const BoundingBox3f &getBoundingBox() const { return m_bbox; }
    bool isEmitter() const { return m_emitter != nullptr; }
    Emitter *getEmitter() { return m_emitter; }
    const Emitter *getEmitter() const { return m_emitter; }
    const BSDF *getBSDF() const { return m_bsdf; }
    const NormalModifier *getNormalModifier() const { return m_normalModifier; }
    virtual uint32_t getPrimitiveCount() const { return 1; }
    virtual BoundingBox3f getBoundingBox(uint32_t index) const = 0;
    virtual Point3f getCentroid(uint32_t index) const = 0;
    virtual bool isSphere() const { return false; }
protected:
    BSDF *m_bsdf = nullptr; Emitter *m_emitter = nullptr; NormalModifier *m_normalModifier = nullptr; BoundingBox3f m_bbox;
};

/// Indexed triangle mesh, struct-of-arrays (include/nori/mesh.h)
class Mesh : public Shape {
public:
    void activate() override;                                // area CDF, src/mesh.cpp:30-38
    uint32_t getPrimitiveCount() const override { return (uint32_t) (m_F.size() / 3); }
    BoundingBox3f getBoundingBox(uint32_t index) const override;
    Point3f getCentroid(uint32_t index) const override;
    float surfaceArea(uint32_t index) const;
    const std::vector<float> &getVertexPositions() const { return m_V; }     // 3 per vertex
    const std::vector<float> &getVertexNormals() const { return m_N; }
    const std::vector<float> &getVertexTexCoords() const { return m_UV; }    // 2 per vertex
    const std::vector<uint32_t> &getIndices() const { return m_F; }          // 3 per triangle
    const std::vector<float> &getAreaCdf() const { return m_cdf; }           // F+1 entries, normalised
    float getAreaNormalization() const { return m_cdfNorm; }
    const std::string &getName() const { return m_name; }
    std::string toString() const override;
    Point3f vertex(uint32_t i) const { return Point3f(m_V[3 * i], m_V[3 * i + 1], m_V[3 * i + 2]); }
protected:
    std::string m_name;
    std::vector<float> m_V, m_N, m_UV; std::vector<uint32_t> m_F;
    std::vector<float> m_cdf; float m_cdfNorm = 0;
};

class Sphere : public Shape {
public:
    explicit Sphere(const PropertyList &props);
    BoundingBox3f getBoundingBox(uint32_t) const override { return m_bbox; }
    Point3f getCentroid(uint32_t) const override { return m_position; }
    bool isSphere() const override { return true; }
    Point3f getCenter() const { return m_position; }
    float getRadius() const { return m_radius; }
    std::string toString() const override;
private:
    Point3f m_position; float m_radius;
};

class ReconstructionFilter : public NoriObject {
public:
    EClassType getClassType() const override { return EReconstructionFilter; }
    float getRadius() const { return m_radius; }
    virtual float eval(float x) const = 0;                   // include/nori/rfilter.h:45
protected:
    float m_radius = 2.0f;
};

class Camera : public NoriObject {
public:
    EClassType getClassType() const override { return ECamera; }
    int getWidth() const { return m_width; }
    int getHeight() const { return m_height; }
    const ReconstructionFilter *getReconstructionFilter() const { return m_rfilter; }
    virtual void describe(NoriHipCamera &out) const = 0;
    void setOutputSize(int w, int h) { m_width = w; m_height = h; activate(); }
protected:
    int m_width = 1280, m_height = 720; ReconstructionFilter *m_rfilter = nullptr;
};

class Sampler : public NoriObject {
public:
    EClassType getClassType() const override { return ESampler; }
    size_t getSampleCount() const { return m_sampleCount; }
    void setSampleCount(size_t n) { m_sampleCount = n; }
protected:
    size_t m_sampleCount = 1;
};

/// SAH BVH with the reference's topology and leaf order (src/bvh.cpp), flattened for the GPU.
class BVH {
public:
    struct Node32 { uint32_t w0, w1; float bmin[3], bmax[3]; };    // information of include/nori/bvh.h:127-164
    ~BVH();
    void addShape(Shape *shape);
    void build(int threads = 0);
    bool isBuilt() const { return !m_nodes.empty(); }
    uint32_t getPrimitiveCount() const { return m_shapeOffset.back(); }
    const std::vector<Shape *> &getShapes() const { return m_shapes; }
    const std::vector<uint32_t> &getShapeOffsets() const { return m_shapeOffset; }
    const BoundingBox3f &getBoundingBox() const { return m_bbox; }
    const std::vector<Node32> &getNodes() const { return m_nodes; }          // reference numbering (pre-order, compacted)
    const std::vector<uint32_t> &getIndices() const { return m_indices; }    // the reference's m_indices
    uint32_t findShape(uint32_t &idx) const;                                 // include/nori/bvh.h:105-109
    double getBuildSeconds() const { return m_buildSeconds; }
    void setOwnsShapes(bool v) { m_owns = v; }
private:
    std::vector<Shape *> m_shapes; std::vector<uint32_t> m_shapeOffset{0u};
    std::vector<Node32> m_nodes; std::vector<uint32_t> m_indices; BoundingBox3f m_bbox; double m_buildSeconds = 0; bool m_owns = true;
};

/// <enviromentMap> (cuda/include/nori/enviromentmap.h): an emitter around the scene
class EnvironmentMap : public NoriObject {
public:
    EClassType getClassType() const override { return EEnvMap; }
    virtual const ImageData *image() const = 0;
    virtual void describe(NoriHipEnvMap &out) const = 0;     // brightness and frame; the texture index is the flattener's
};

class Scene;
/// Integrator plugin interface (include/nori/integrator.h:35-62).  The reference's per-ray Li() cannot feed a GPU;
/// the batch entry render() is what the (headless) render driver calls after preprocess().
class Integrator : public NoriObject {
public:
    EClassType getClassType() const override { return EIntegrator; }
    virtual void preprocess(const Scene *) {}
    /// true: the integrator builds the acceleration structure itself (on the device); Scene::activate then skips the host SAH build
    virtual bool buildsOwnBVH() const { return false; }
    /// Radiance along one ray (include/nori/integrator.h:48-58).  The GPU integrators evaluate it as a batch of one - correct,
    /// and as slow as one kernel launch per ray: render() / nori_hip_li_samples are the interfaces for throughput.
    virtual Color3f Li(const Scene *scene, Sampler *sampler, const Ray3f &ray) const;
    /// Accumulates samples [spp_begin, spp_end) of every pixel into the integrator's film
    virtual void render(const Scene *scene, int spp_begin, int spp_end) = 0;
    /// rows x cols x 4 float film (ImageBlock layout) and its border
    virtual void readFilm(std::vector<float> &film, int &rows, int &cols, int &border) = 0;
    /// height x width x 3, rgb / weight (ImageBlock::toBitmap)
    virtual void readBitmap(std::vector<float> &rgb) = 0;
};

class Scene : public NoriObject {
public:
    explicit Scene(const PropertyList &);
    ~Scene() override;
    EClassType getClassType() const override { return EScene; }
    void addChild(NoriObject *obj) override;                 // src/scene.cpp:62-99
    void activate() override;                                // builds the BVH, src/scene.cpp:42-60
    const BVH *getBVH() const { return m_bvh; }
    const Integrator *getIntegrator() const { return m_integrator; }
    Integrator *getIntegrator() { return m_integrator; }
    const Camera *getCamera() const { return m_camera; }
    Camera *getCamera() { return m_camera; }
    const Sampler *getSampler() const { return m_sampler; }
    Sampler *getSampler() { return m_sampler; }
    const std::vector<Shape *> &getShapes() const { return m_shapes; }
    const std::vector<Emitter *> &getLights() const { return m_emitters; }
    const EnvironmentMap *getEnvMap() const { return m_envmap; }          // cuda/include/nori/scene.h:79
    std::string toString() const override;
private:
    std::vector<Shape *> m_shapes; std::vector<Emitter *> m_emitters;
    Integrator *m_integrator = nullptr; Sampler *m_sampler = nullptr; Camera *m_camera = nullptr; BVH *m_bvh = nullptr; EnvironmentMap *m_envmap = nullptr;
};

/// Owns the arrays a NoriHipSceneDesc points into
struct FlatScene {
    std::vector<NoriHipNode> nodes; std::vector<NoriHipPrim> prims; std::vector<NoriHipShadePrim> shade, light_prims;
    std::vector<NoriHipShape> shapes; std::vector<NoriHipEmitter> emitters; std::vector<float> cdf;
    std::vector<NoriHipTexture> textures; std::vector<uint8_t> tex_data;
    NoriHipSceneDesc desc;
};
/// Scene graph -> C ABI tables (replaces the object-by-object marshalling of cuda/CudaRender.cpp:80-98)
void flattenScene(const Scene *scene, FlatScene &out);

} // namespace nori
