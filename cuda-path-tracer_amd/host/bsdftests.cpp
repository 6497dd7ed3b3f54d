// The BSDF modes of Nori's statistical test plugins, run on the device's BSDF code through nori_hip_bsdf_vectors:
//   <test type="chi2test">  histogram of BSDF::sample() against the integral of BSDF::pdf() on a (cos theta, phi) grid
//                           (src/chi2test.cpp:93-196, ext/hypothesis/hypothesis.h:140-235)
//   <test type="ttest"> with <bsdf> children: mean luminance of the sample weights at given angles (src/ttest.cpp:107-147)
// The statistics (histogram, pooling of sparse cells, chi^2 and t distributions, Sidak correction) are evaluated on the host as
// in the reference; every BSDF value comes from the GPU.  The random numbers are the reference's: one default-seeded pcg32
// carried across all tests (src/chi2test.cpp:96, src/ttest.cpp:104), four floats per sample.
#include "render.h"
#include <cstring>
#include <iostream>

namespace nori {

namespace {

// ext/pcg32/pcg32.h:51-110 (host copy: the test plugins draw their sample positions on the host, as the reference does)
struct HostPcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t nextUInt() { uint64_t o = state; state = o * 0x5851f42d4c957f2dULL + inc; uint32_t x = (uint32_t) (((o >> 18u) ^ o) >> 27u), r = (uint32_t) (o >> 59u); return (x >> r) | (x << ((~r + 1u) & 31)); }
    float nextFloat() { union { uint32_t u; float f; } x; x.u = (nextUInt() >> 9) | 0x3f800000u; return x.f - 1.0f; }
};

void hipCheck(int rc, const char *what) { if (rc != NORI_HIP_OK) throw NoriException::format("%s failed: %s", what, nori_hip_last_error()); }

// regularised lower incomplete gamma P(a, x): series for x < a + 1, Lentz's continued fraction otherwise
double gammaP(double a, double x) {
    if (x <= 0) return 0;
    const double gln = std::lgamma(a);
    if (x < a + 1) {
        double ap = a, sum = 1 / a, del = sum;
        for (int n = 0; n < 1000; ++n) { ap += 1; del *= x / ap; sum += del; if (std::fabs(del) < std::fabs(sum) * 1e-16) break; }
        return sum * std::exp(-x + a * std::log(x) - gln);
    }
    const double tiny = 1e-300; double b = x + 1 - a, c = 1 / tiny, d = 1 / b, h = d;
    for (int i = 1; i < 1000; ++i) {
        double an = -i * (i - a); b += 2; d = an * d + b; if (std::fabs(d) < tiny) d = tiny; c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1 / d; double del = d * c; h *= del; if (std::fabs(del - 1) < 1e-16) break;
    }
    return 1 - std::exp(-x + a * std::log(x) - gln) * h;
}

NoriHipContext *testContext() {
    static NoriHipContext *ctx = nullptr;                                 // one context for all BSDF tests of the process (never destroyed)
    if (!ctx) { const char *env = getenv("NORI_HIP_DEVICE"); hipCheck(nori_hip_create(env ? atoi(env) : 0, &ctx), "nori_hip_create"); }
    return ctx;
}

NoriHipShape describeBsdf(const BSDF *bsdf) {
    NoriHipShape s; std::memset(&s, 0, sizeof(s)); s.emitter = -1;
    bsdf->describe(s);
    if (bsdf->albedoImage()) throw NoriException("BSDF tests evaluate constant-colour BSDFs");
    return s;
}

} // namespace

double chi2Cdf(double x, int dof) { return gammaP(0.5 * dof, 0.5 * x); }

// hypothesis::chi2_test (ext/hypothesis/hypothesis.h:140-235)
Chi2Result chi2Test(int nCells, const double *obs, const double *expd, int sampleCount, double minExpFrequency, double significanceLevel, int numTests) {
    Chi2Result r;
    std::vector<int> order(nCells); for (int i = 0; i < nCells; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return expd[a] < expd[b]; });
    double pooledObs = 0, pooledExp = 0, chsq = 0; int dof = 0;
    for (int idx : order) {
        if (expd[idx] == 0) {
            if (obs[idx] > sampleCount * 1e-5) { r.message = "samples in a cell with expected frequency 0"; return r; }
        } else if (expd[idx] < minExpFrequency) { pooledObs += obs[idx]; pooledExp += expd[idx]; r.pooledCells++; }
        else if (pooledExp > 0 && pooledExp < minExpFrequency) { pooledObs += obs[idx]; pooledExp += expd[idx]; r.pooledCells++; }
        else { double diff = obs[idx] - expd[idx]; chsq += (diff * diff) / expd[idx]; ++dof; }
    }
    if (pooledExp > 0 || pooledObs > 0) { double diff = pooledObs - pooledExp; chsq += (diff * diff) / pooledExp; ++dof; }
    dof -= 1;
    r.chsq = chsq; r.dof = dof;
    if (dof <= 0) { r.message = "the number of degrees of freedom is too low"; return r; }
    r.pvalue = 1 - chi2Cdf(chsq, dof);
    r.alpha = 1.0 - std::pow(1.0 - significanceLevel, 1.0 / numTests);
    r.passed = !(r.pvalue < r.alpha || !std::isfinite(r.pvalue));
    return r;
}

ChiSquareTest::ChiSquareTest(const PropertyList &p) {
    m_significanceLevel = p.getFloat("significanceLevel", 0.01f);
    m_cosThetaResolution = p.getInteger("resolution", 10);
    m_minExpFrequency = p.getInteger("minExpFrequency", 5);
    m_sampleCount = p.getInteger("sampleCount", -1);
    m_testCount = p.getInteger("testCount", 5);
    m_phiResolution = 2 * m_cosThetaResolution;
    if (m_sampleCount < 0) m_sampleCount = m_cosThetaResolution * m_phiResolution * 5000;       // ~5K samples per bin
}
ChiSquareTest::~ChiSquareTest() { for (auto b : m_bsdfs) delete b; }
void ChiSquareTest::addChild(NoriObject *obj) {
    if (obj->getClassType() != EBSDF) throw NoriException::format("ChiSquareTest::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()).c_str());
    m_bsdfs.push_back(static_cast<BSDF *>(obj));
}
std::string ChiSquareTest::toString() const {
    char b[256]; snprintf(b, sizeof(b), "ChiSquareTest[\n  thetaResolution = %i,\n  phiResolution = %i,\n  minExpFrequency = %i,\n  sampleCount = %i,\n  testCount = %i,\n  significanceLevel = %f\n]",
                          m_cosThetaResolution, m_phiResolution, m_minExpFrequency, m_sampleCount, m_testCount, m_significanceLevel);
    return b;
}
bool ChiSquareTest::s_runOnActivate = true;

// All tests of the file, in the reference's order, with its one sampler carried through (src/chi2test.cpp:93-196)
std::vector<Chi2Result> ChiSquareTest::run() {
    NoriHipContext *ctx = testContext();
    const int res = m_cosThetaResolution * m_phiResolution, n = m_sampleCount;
    HostPcg32 random;
    std::vector<Chi2Result> results;
    std::vector<float> wi((size_t) 3 * n), s1((size_t) 2 * n), s2((size_t) 2 * n), out((size_t) 12 * n);
    // the expected frequencies: pdf integrated over every cell with Simpson's rule on a fixed (2 K + 1)^2 grid per cell (the
    // reference refines adaptively, ext/hypothesis/hypothesis.h:60-126; K = 16 halves resolve the Beckmann lobes of the test files)
    const int K = 16, M = 2 * K + 1;
    std::vector<float> gwi((size_t) 3 * res * M * M), gwo((size_t) 3 * res * M * M), gout((size_t) 12 * res * M * M);
    for (const BSDF *bsdf : m_bsdfs) {
        const NoriHipShape shape = describeBsdf(bsdf);
        for (int l = 0; l < m_testCount; ++l) {
            std::vector<double> obs(res, 0.0), expd(res, 0.0);
            const float cosTheta = random.nextFloat();
            const float sinTheta = std::sqrt(std::max(0.0f, 1 - cosTheta * cosTheta));
            float sinPhi, cosPhi; sincosf(2.0f * 3.14159265358979323846f * random.nextFloat(), &sinPhi, &cosPhi);
            const float w[3] = {cosPhi * sinTheta, sinPhi * sinTheta, cosTheta};
            for (int i = 0; i < n; ++i) {
                wi[3 * i] = w[0]; wi[3 * i + 1] = w[1]; wi[3 * i + 2] = w[2];
                s1[2 * i] = random.nextFloat(); s1[2 * i + 1] = random.nextFloat(); s2[2 * i] = random.nextFloat(); s2[2 * i + 1] = random.nextFloat();
            }
            hipCheck(nori_hip_bsdf_vectors(ctx, &shape, (size_t) n, wi.data(), s1.data(), s2.data(), nullptr, out.data()), "nori_hip_bsdf_vectors");
            for (int i = 0; i < n; ++i) {
                const float *o = &out[12 * (size_t) i];
                if (o[0] == 0 && o[1] == 0 && o[2] == 0) continue;
                int cb = std::min(std::max(0, (int) std::floor((o[5] * 0.5f + 0.5f) * m_cosThetaResolution)), m_cosThetaResolution - 1);
                float scaledPhi = std::atan2(o[4], o[3]) * (0.5f / 3.14159265358979323846f);
                if (scaledPhi < 0) scaledPhi += 1;
                int pb = std::min(std::max(0, (int) std::floor(scaledPhi * m_phiResolution)), m_phiResolution - 1);
                obs[cb * m_phiResolution + pb] += 1;
            }
            size_t q = 0;
            for (int i = 0; i < m_cosThetaResolution; ++i) for (int j = 0; j < m_phiResolution; ++j)
                for (int a = 0; a < M; ++a) for (int b = 0; b < M; ++b, ++q) {
                    const double ct = -1.0 + (i + a / (double) (M - 1)) * 2.0 / m_cosThetaResolution, ph = (j + b / (double) (M - 1)) * 2 * M_PI / m_phiResolution;
                    const double st = std::sqrt(std::max(0.0, 1 - ct * ct));
                    gwo[3 * q] = (float) (st * std::cos(ph)); gwo[3 * q + 1] = (float) (st * std::sin(ph)); gwo[3 * q + 2] = (float) ct;
                    gwi[3 * q] = w[0]; gwi[3 * q + 1] = w[1]; gwi[3 * q + 2] = w[2];
                }
            hipCheck(nori_hip_bsdf_vectors(ctx, &shape, q, gwi.data(), nullptr, nullptr, gwo.data(), gout.data()), "nori_hip_bsdf_vectors");
            q = 0;
            const double ha = (2.0 / m_cosThetaResolution) / (M - 1), hb = (2 * M_PI / m_phiResolution) / (M - 1);
            for (int c = 0; c < res; ++c) {
                double sum = 0;
                for (int a = 0; a < M; ++a) for (int b = 0; b < M; ++b, ++q) {
                    const double wa = (a == 0 || a == M - 1) ? 1 : (a & 1) ? 4 : 2, wb = (b == 0 || b == M - 1) ? 1 : (b & 1) ? 4 : 2;
                    sum += wa * wb * (double) gout[12 * q + 6];
                }
                expd[c] = sum * ha * hb / 9.0 * n;
            }
            results.push_back(chi2Test(res, obs.data(), expd.data(), n, m_minExpFrequency, m_significanceLevel, m_testCount * (int) m_bsdfs.size()));
        }
    }
    return results;
}
void ChiSquareTest::activate() {
    if (!s_runOnActivate) return;
    int passed = 0; m_results = run();
    for (const Chi2Result &r : m_results) {
        std::cout << "------------------------------------------------------\n";
        if (!r.message.empty()) std::cout << r.message << ". Rejecting the null hypothesis!\n";
        else std::cout << "Chi^2 statistic = " << r.chsq << " (d.o.f. = " << r.dof << ")\n" << (r.passed ? "Accepted" : "***** Rejected *****")
                       << " the null hypothesis (p-value = " << r.pvalue << ", significance level = " << r.alpha << ")" << std::endl;
        if (r.passed) ++passed;
    }
    std::cout << "Passed " << passed << "/" << m_results.size() << " tests." << std::endl;
}
NORI_REGISTER_CLASS(ChiSquareTest, "chi2test")

// ---- <test type="ttest"> with BSDF children (src/ttest.cpp:107-147): all tests of the file in order, one sampler
std::vector<TTestResult> StudentsTTest::runBsdfs() {
    if (m_references.size() != m_angles.size() * m_bsdfs.size() && m_references.size() * m_bsdfs.size() != m_angles.size())
        throw NoriException("Specified a different number of angles and reference values!");
    if (!m_scenes.empty()) throw NoriException("Cannot test BSDFs and scenes at the same time!");
    NoriHipContext *ctx = testContext();
    HostPcg32 random;
    const int n = m_sampleCount;
    std::vector<float> wi((size_t) 3 * n), s1((size_t) 2 * n), s2((size_t) 2 * n), out((size_t) 12 * n);
    std::vector<TTestResult> results;
    // (the reference indexes angles by the inner counter and references by a running one, and requires
    // references x bsdfs == angles, which only holds for one BSDF per file; here every BSDF is tested at every angle and
    // reference k belongs to test k)
    size_t ctr = 0;
    for (const BSDF *bsdf : m_bsdfs) {
        const NoriHipShape shape = describeBsdf(bsdf);
        for (size_t a = 0; a < m_angles.size() && ctr < m_references.size(); ++a) {
            const float theta = degToRad(m_angles[a]);
            const float w[3] = {std::sin(theta), 0.0f, std::cos(theta)};                    // sphericalDirection(theta, 0)
            for (int i = 0; i < n; ++i) {
                wi[3 * i] = w[0]; wi[3 * i + 1] = w[1]; wi[3 * i + 2] = w[2];
                s1[2 * i] = random.nextFloat(); s1[2 * i + 1] = random.nextFloat(); s2[2 * i] = random.nextFloat(); s2[2 * i + 1] = random.nextFloat();
            }
            hipCheck(nori_hip_bsdf_vectors(ctx, &shape, (size_t) n, wi.data(), s1.data(), s2.data(), nullptr, out.data()), "nori_hip_bsdf_vectors");
            double mean = 0, variance = 0;
            for (int k = 0; k < n; ++k) {
                const float *o = &out[12 * (size_t) k];
                double result = (double) (o[0] * 0.212671f + o[1] * 0.715160f + o[2] * 0.072169f);
                double delta = result - mean; mean += delta / (double) (k + 1); variance += delta * (result - mean);
            }
            variance /= n - 1;
            results.push_back(studentsTTest(mean, variance, m_references[ctr++], n, m_significanceLevel, (int) m_references.size()));
        }
    }
    return results;
}

} // namespace nori
