// Declarations shared by the integrator plugin, the t-test plugin, the render driver and the host C API.
#pragma once
#include "scene.h"

namespace nori {

/// An Integrator whose work is done by a NoriHipContext
class GpuIntegrator : public Integrator {
public:
    virtual NoriHipContext *context() = 0;
    virtual const FlatScene &flat() const = 0;
};

struct TTestResult { double mean = 0, variance = 0, reference = 0, t = 0, pvalue = 0, alpha = 0; bool passed = false; };
double studentsTCdf(double t, double dof);
/// ext/hypothesis/hypothesis.h:313-346
TTestResult studentsTTest(double mean, double variance, double reference, int sampleCount, double significanceLevel, int numTests);

/// <test type="ttest"> in scene mode (src/ttest.cpp:148-195), run through the GPU integrator
class StudentsTTest : public NoriObject {
public:
    explicit StudentsTTest(const PropertyList &propList);
    ~StudentsTTest() override;
    EClassType getClassType() const override { return ETest; }
    void addChild(NoriObject *obj) override;
    void activate() override;
    std::string toString() const override;
    TTestResult runScene(size_t i);
    /// BSDF mode (src/ttest.cpp:107-147): every <bsdf> child at every angle, on the device's BSDF code (host/bsdftests.cpp)
    std::vector<TTestResult> runBsdfs();
    bool bsdfMode() const { return !m_bsdfs.empty(); }
    const std::vector<Scene *> &scenes() const { return m_scenes; }
    const std::vector<float> &references() const { return m_references; }
    const std::vector<TTestResult> &results() const { return m_results; }
    int sampleCount() const { return m_sampleCount; }
    float significanceLevel() const { return m_significanceLevel; }
    static bool s_runOnActivate;           // the reference always runs inside the parser; tools that only want the scenes clear this
private:
    std::vector<BSDF *> m_bsdfs; std::vector<Scene *> m_scenes; std::vector<float> m_angles, m_references; std::vector<TTestResult> m_results;
    float m_significanceLevel; int m_sampleCount;
};

struct Chi2Result { double chsq = 0, pvalue = 0, alpha = 0; int dof = 0, pooledCells = 0; bool passed = false; std::string message; };
double chi2Cdf(double x, int dof);
/// ext/hypothesis/hypothesis.h:140-235
Chi2Result chi2Test(int nCells, const double *obsFrequencies, const double *expFrequencies, int sampleCount, double minExpFrequency, double significanceLevel, int numTests);

/// <test type="chi2test"> (src/chi2test.cpp): BSDF::sample against BSDF::pdf, both evaluated on the GPU (nori_hip_bsdf_vectors)
class ChiSquareTest : public NoriObject {
public:
    explicit ChiSquareTest(const PropertyList &propList);
    ~ChiSquareTest() override;
    EClassType getClassType() const override { return ETest; }
    void addChild(NoriObject *obj) override;
    void activate() override;
    std::string toString() const override;
    std::vector<Chi2Result> run();
    const std::vector<Chi2Result> &results() const { return m_results; }
    static bool s_runOnActivate;
private:
    int m_cosThetaResolution, m_phiResolution, m_minExpFrequency, m_sampleCount, m_testCount; float m_significanceLevel;
    std::vector<BSDF *> m_bsdfs; std::vector<Chi2Result> m_results;
};

/// Loads, renders and writes <scene>.pfm next to the XML; returns the output path (src/render.cpp:103-214)
std::string renderScene(const std::string &filename, int passSize = 0);
void writePFM(const std::string &path, const std::vector<float> &rgb, int w, int h);
void writePNG(const std::string &path, const std::vector<float> &rgb, int w, int h);
void writeEXR(const std::string &path, const std::vector<float> &rgb, int w, int h);
void arrayMeshSet(Mesh *m, const std::string &name, const float *V, size_t nV, const uint32_t *F, size_t nF, const float *N, const float *UV);

} // namespace nori
