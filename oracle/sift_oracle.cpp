// TEST INFRASTRUCTURE ONLY (see or_common.h) — CPU restatement of the SiftGPU fork used by the
// reference: Gaussian pyramid, DoG extrema gated by valid depth, <=2 orientations, 4x4x8 descriptor.
//   SiftGPU/ProgramCU.cu:159-264 (separable filters), :330-354 (down-sample), :431-460 (filter taps),
//   :550-582 (DoG + gradient), :616-750 (keypoints), :905-1142 (orientation), :1178-1257 (descriptor),
//   :1339-1370 (normalise), :1994-2107 (reshape, key list, uchar);  SiftGPU/SiftPyramid.cpp:82-145,
//   148-255, 297-314, 351-394, 426-450, 730-768;  SiftGPU/SiftGPU.cpp:105-174, 224-253.
// PINNED to the reference's SiftGPU fork through oracle/_ref (tests/test_ref_pin_cpu.py::test_sift_detector_and_matcher_vs_reference): all 18
// pyramid levels and the DoG extrema sets bit for bit, per-slot counts, the final key points as a multiset of float bits, descriptors within one
// count in <= 0.1 % of the bytes (the reference build uses CUDA's fast-math exp / atan2 / sincos, the host build of it glibc, this file the fixed
// sequences of include/bf_detmath.h).  Canonical choices where the reference is order-dependent:
//   * keypoints of a level are kept in (row, col) order (the reference appends with atomicAdd);
//   * the orientation histogram and the descriptor bins are accumulated as 64 strided partial sums
//     (sample s goes to partial s mod 64) combined by a xor-butterfly 32,16,...,1 — the summation tree
//     of a 64-wide wavefront (the reference uses shared-memory float atomics in arbitrary order);
//   * exp / atan2 / sincos come from include/bf_detmath.h (the reference uses fast-math intrinsics).
#include <algorithm>
#include <cstdio>
#include <vector>

#include "../include/bf_detmath.h"
#include "../include/bf_hip.h"
#include "or_common.h"

using namespace orc;

namespace {

const int NUM_OCT = 4, DOG_LEVELS = 3, LEVEL_MIN = -1, LEVEL_MAX = 4, NLEV = 6;
const float PI_F = 3.14159265358979323846f;

struct SiftParams {
    float sigma[5];            // incremental blur of levels 0..4
    float sigma0;
    float initSigma;           // blur input -> level -1
    float dogThreshold, edgeThreshold;
    std::vector<float> taps[6];
};

void makeTaps(float sigma, std::vector<float>& k) {     // CreateFilterKernel :431-460
    int sz = (int)ceil(4.0f * sigma - 0.5);
    int width = 2 * sz + 1;
    if (width > 33) { sz = 16; width = 33; } else if (width < 5) { sz = 2; width = 5; }
    k.resize(width);
    float rv = 1.0f / (sigma * sigma), ksum = 0;
    for (int i = -sz; i <= sz; ++i) { float v = expf(-0.5f * i * i * rv); k[i + sz] = v; ksum += v; }
    rv = 1.0f / ksum;
    for (int i = 0; i < width; ++i) k[i] *= rv;
}

SiftParams makeParams() {       // SiftParam::ParseSiftParam, SiftGPU.cpp:126-174
    SiftParams p;
    p.sigma0 = 1.6f * powf(2.0f, 1.0f / DOG_LEVELS);
    const float sigmak = powf(2.0f, 1.0f / DOG_LEVELS);
    const float dsigma0 = p.sigma0 * sqrtf(1.0f - 1.0f / (sigmak * sigmak));
    for (int i = LEVEL_MIN + 1; i <= LEVEL_MAX; ++i) p.sigma[i - LEVEL_MIN - 1] = dsigma0 * powf(sigmak, (float)i);
    const float sa = p.sigma0 * powf(2.0f, (float)LEVEL_MIN / (float)DOG_LEVELS), sb = 0.5f;
    p.initSigma = sa > sb + 0.001 ? sqrtf(sa * sa - sb * sb) : 0.0f;
    p.dogThreshold = 0.02f / DOG_LEVELS;
    p.edgeThreshold = 10.0f;
    makeTaps(p.initSigma, p.taps[0]);
    for (int i = 0; i < 5; ++i) makeTaps(p.sigma[i], p.taps[i + 1]);
    return p;
}

void blur(const std::vector<float>& src, std::vector<float>& dst, int w, int h, const std::vector<float>& k) {
    const int fw = (int)k.size(), half = fw >> 1;
    std::vector<float> tmp((size_t)w * h);
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (int y = 0; y < h; ++y)          // FilterH :159-196 (clamp to edge)
        for (int x = 0; x < w; ++x) {
            float v = 0;
            for (int i = 0; i < fw; ++i) { int xx = std::min(std::max(x - half + i, 0), w - 1); v += src[(size_t)y * w + xx] * k[i]; }
            tmp[(size_t)y * w + x] = v;
        }
    dst.resize((size_t)w * h);
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (int y = 0; y < h; ++y)          // FilterV :198-264
        for (int x = 0; x < w; ++x) {
            float v = 0;
            for (int i = 0; i < fw; ++i) { int yy = std::min(std::max(y - half + i, 0), h - 1); v += tmp[(size_t)yy * w + x] * k[i]; }
            dst[(size_t)y * w + x] = v;
        }
}

struct Level { std::vector<float> g; std::vector<float> mag, ang; };   // gaussian, gradient magnitude / angle
struct Octave { int w, h; Level lev[NLEV]; };

inline float fetchLin(const std::vector<float>& a, long idx) { return (idx < 0 || idx >= (long)a.size()) ? 0.0f : a[idx]; }   // tex1Dfetch out of range = 0

struct RawKey { int x, y, level; float ori[2]; int nOri; };

inline float butterfly(float* lane) {       // 64-lane xor butterfly sum
    for (int o = 32; o > 0; o >>= 1) {
        float t[64];
        for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ o];
        memcpy(lane, t, sizeof t);
    }
    return lane[0];
}

}  // namespace

// BuildPyramid + DetectKeypoints (SiftPyramid.cpp:82-145, 351-394): the pyramid and the raw key lists per (octave, DoG level) slot
static void buildAndDetect(const SiftParams& P, const float* intensity, const float* depth, int W, int H, int depthW, int depthH, float depthMin, float depthMax,
                           std::vector<Octave>& oct, std::vector<RawKey>* raw, int* levelNum) {
    // ---- BuildPyramid, SiftPyramid.cpp:82-145
    for (int o = 0; o < NUM_OCT; ++o) {
        Octave& oc = oct[o];
        oc.w = W >> o; oc.h = H >> o;
        if (o == 0) {
            std::vector<float> in(intensity, intensity + (size_t)W * H);
            blur(in, oc.lev[0].g, oc.w, oc.h, P.taps[0]);
        } else {
            const Octave& pv = oct[o - 1];
            const std::vector<float>& src = pv.lev[3].g;     // level_ds - level_min = 3  (level 2)
            oc.lev[0].g.resize((size_t)oc.w * oc.h);
            for (int y = 0; y < oc.h; ++y)                   // DownsampleKernel :330-354
                for (int x = 0; x < oc.w; ++x) oc.lev[0].g[(size_t)y * oc.w + x] = src[(size_t)(y << 1) * pv.w + std::min(x << 1, pv.w - 1)];
        }
        for (int j = 0; j <= LEVEL_MAX; ++j) blur(oc.lev[j].g, oc.lev[j + 1].g, oc.w, oc.h, P.taps[j + 1]);
        // gradient of gaussian levels 0..2 (array index 1..3), ComputeDOG_Kernel :550-569
        for (int a = 1; a <= 3; ++a) {
            Level& L = oc.lev[a];
            L.mag.resize(L.g.size()); L.ang.resize(L.g.size());
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
            for (int y = 0; y < oc.h; ++y)
                for (int x = 0; x < oc.w; ++x) {
                    const long idx = (long)y * oc.w + x;
                    const float dx = fetchLin(L.g, idx + 1) - fetchLin(L.g, idx - 1);
                    const float dy = fetchLin(L.g, idx + oc.w) - fetchLin(L.g, idx - oc.w);
                    const float grd = 0.5f * sqrtf(dx * dx + dy * dy);
                    L.mag[idx] = grd;
                    L.ang[idx] = grd == 0.0f ? 0.0f : bf_dm_atan2(dy, dx);
                }
        }
    }
    // ---- DetectKeypoints, SiftPyramid.cpp:351-394 + ComputeKEY_Kernel :616-750
    const float Tedge = (P.edgeThreshold + 1) * (P.edgeThreshold + 1) / P.edgeThreshold;
    for (int o = 0; o < NUM_OCT; ++o) {
        const Octave& oc = oct[o];
        const int w = oc.w, h = oc.h;
        const float keyLocScale = (float)(1 << o), keyLocOffset = 0.5f;
        for (int j = 1; j <= 3; ++j) {           // DoG level j = g[j+1] - g[j]; prev j-1, next j+1
            const int li = o * DOG_LEVELS + j - 1;
            auto dog = [&](int lev, long idx) -> float {      // tex1Dfetch of the DoG image of level `lev`
                if (idx < 0 || idx >= (long)w * h) return 0.0f;
                return oc.lev[lev + 1].g[idx] - oc.lev[lev].g[idx];
            };
            const int fmaxRaw = (int)(w * h * 0.005f);
            const int fmax = fmaxRaw > 4096 ? 4096 : (fmaxRaw < 32 ? 32 : fmaxRaw);
            for (int row = 1; row < h - 2; ++row)
                for (int col = 1; col < w - 2; ++col) {
                    const long index = (long)row * w + col;
                    const int depthx = f2i(roundf((keyLocScale * (float)col + keyLocOffset) * (float)(depthW - 1) / (float)(W - 1)));
                    const int depthy = f2i(roundf((keyLocScale * (float)row + keyLocOffset) * (float)(depthH - 1) / (float)(H - 1)));
                    if (depthx < 0 || depthx >= depthW || depthy < 0 || depthy >= depthH) continue;
                    const float dd = depth[(size_t)depthy * depthW + depthx];
                    if (dd == MINF || dd < depthMin || dd > depthMax) continue;
                    const float v = dog(j, index);
                    if (fabsf(v) <= P.dogThreshold) continue;
                    float d[3][3];
                    d[1][1] = v; d[1][0] = dog(j, index - 1); d[1][2] = dog(j, index + 1);
                    float nmax = std::max(d[1][0], d[1][2]), nmin = std::min(d[1][0], d[1][2]);
                    if (v <= nmax && v >= nmin) continue;
                    bool reject = false;
                    auto cmp3 = [&](float* out, int lev, long idx) {      // READ_CMP_DOG_DATA :596-614
                        out[0] = dog(lev, idx - 1); out[1] = dog(lev, idx); out[2] = dog(lev, idx + 1);
                        if (v > nmax) { nmax = std::max(nmax, out[0]); nmax = std::max(nmax, out[1]); nmax = std::max(nmax, out[2]); if (v < nmax) reject = true; }
                        else { nmin = std::min(nmin, out[0]); nmin = std::min(nmin, out[1]); nmin = std::min(nmin, out[2]); if (v > nmin) reject = true; }
                    };
                    cmp3(d[0], j, index - w); if (reject) continue;
                    cmp3(d[2], j, index + w); if (reject) continue;
                    const float vx2 = v * 2.0f;
                    const float fxx = d[1][0] + d[1][2] - vx2, fyy = d[0][1] + d[2][1] - vx2;
                    const float fxy = 0.25f * (d[2][2] + d[0][0] - d[2][0] - d[0][2]);
                    const float t1 = fxx * fyy - fxy * fxy, t2 = (fxx + fyy) * (fxx + fyy);
                    if (t1 <= 0 || t2 > Tedge * t1) continue;
                    float tmp[3];
                    cmp3(tmp, j - 1, index - w); if (reject) continue;
                    cmp3(tmp, j - 1, index); if (reject) continue;
                    cmp3(tmp, j - 1, index + w); if (reject) continue;
                    cmp3(tmp, j + 1, index - w); if (reject) continue;
                    cmp3(tmp, j + 1, index); if (reject) continue;
                    cmp3(tmp, j + 1, index + w); if (reject) continue;
                    if ((int)raw[li].size() < fmax) raw[li].push_back({col, row, li, {0, 0}, 0});
                }
            levelNum[li] = (int)raw[li].size();
        }
    }
}

extern "C" {

// returns the number of features; keys: 4 floats each (x, y, scale, depth); descs: 128 bytes each.
// levelCounts (optional, 12 ints): per (octave, level) feature count after the final limit.
int or_sift_run(const float* intensity, const float* depth, int W, int H, int depthW, int depthH, float depthMin, float depthMax,
                float minKeyScale, int featureCountThreshold, int maxFeatures, float* keys, uint8_t* descs, int* levelCounts) {
    static SiftParams P = makeParams();
    std::vector<Octave> oct(NUM_OCT);
    std::vector<RawKey> raw[NUM_OCT * DOG_LEVELS];
    int levelNum[NUM_OCT * DOG_LEVELS];
    buildAndDetect(P, intensity, depth, W, H, depthW, depthH, depthMin, depthMax, oct, raw, levelNum);
    // ---- LimitFeatureCount(0), SiftPyramid.cpp:227-255 (TruncateMethod 0)
    auto limit = [&](int* lv) {
        if (featureCountThreshold <= 0) return;
        int total = 0;
        for (int i = 0; i < NUM_OCT * DOG_LEVELS; ++i) total += lv[i];
        int i = 0;
        while (i < NUM_OCT * DOG_LEVELS && total - lv[i] > featureCountThreshold) { total -= lv[i]; lv[i++] = 0; }
    };
    limit(levelNum);
    // ---- GetFeatureOrientations, SiftPyramid.cpp:426-450 + ComputeOrientation_Kernel :905-1142
    for (int li = 0; li < NUM_OCT * DOG_LEVELS; ++li) {
        if (levelNum[li] == 0) { raw[li].clear(); continue; }
        const int o = li / DOG_LEVELS, j = li % DOG_LEVELS;
        const Octave& oc = oct[o];
        const Level& L = oc.lev[j + 1];
        const int width = oc.w, height = oc.h;
        const float sigma = P.sigma0 * powf(2.0f, (float)j / (float)DOG_LEVELS);     // GetLevelSigma(j)
        for (RawKey& k : raw[li]) {
            const float kx = k.x + 0.5f, ky = k.y + 0.5f;
            const float gsigma = sigma * 1.5f;
            const float win = fabsf(sigma) * 1.5f * 2.0f;
            const float dist_threshold = win * win + 0.5f;
            const float factor = -0.5f / (gsigma * gsigma);
            const float xmin = std::max(1.5f, floorf(kx - win) + 0.5f), ymin = std::max(1.5f, floorf(ky - win) + 0.5f);
            const float xmax = std::min(width - 1.5f, floorf(kx + win) + 0.5f), ymax = std::min(height - 1.5f, floorf(ky + win) + 0.5f);
            const unsigned xlen = f2u(roundf(xmax - xmin + 1)), ylen = f2u(roundf(ymax - ymin + 1));
            const unsigned num = xlen * ylen;
            static float part[36][64];
            for (int b = 0; b < 36; ++b) for (int l = 0; l < 64; ++l) part[b][l] = 0.0f;
            for (unsigned s = 0; s < num; ++s) {
                const float x = (float)(s % xlen) + xmin, y = (float)(s / xlen) + ymin;
                const float dx = x - kx, dy = y - ky;
                const float sq = dx * dx + dy * dy;
                if (sq < dist_threshold) {
                    const size_t pi = (size_t)f2i(y) * width + (size_t)f2i(x);
                    const float weight = L.mag[pi] * bf_dm_exp(sq * factor);
                    int oidx = f2i(floorf(L.ang[pi] * 5.7295779513082320876798154814105f));
                    if (oidx < 0) oidx += 36;
                    if (oidx > 35) oidx = 35;
                    part[oidx][s & 63] += weight;
                }
            }
            float vote[36], tmpv[36];
            for (int b = 0; b < 36; ++b) vote[b] = butterfly(part[b]);
            float *src = vote, *dst = tmpv;
            for (int it = 0; it < 6; ++it) {           // :987-1003
                for (int t = 0; t < 36; ++t) dst[t] = (src[(t + 35) % 36] + src[t] + src[(t + 1) % 36]) * (float)(1.0 / 3.0);
                std::swap(src, dst);
            }
            // after 6 swaps src == vote
            float maxv = 0.0f;
            for (int t = 0; t < 36; ++t) maxv = std::max(maxv, vote[t]);
            const float thr = maxv * 0.8f;
            float rot[2] = {0, 0}; int ocount = 0, maxIndex = -1;
            for (int pass = 0; pass < 2; ++pass) {
                float bw = -1.0f; int bi = -1;
                for (int c = 0; c < 36; ++c) {
                    if (pass == 1 && c == maxIndex) continue;
                    if (vote[c] > thr && vote[c] > vote[(c + 35) % 36] && vote[c] > vote[(c + 1) % 36])
                        if (bw < vote[c]) { bw = vote[c]; bi = c; }       // strict: ties keep the lower bin
                }
                if (bi >= 0) {
                    const int m = (bi + 35) % 36, p = (bi + 1) % 36;
                    const float di = 0.5f * ((vote[p] - vote[m]) / (2.0f * vote[bi] - vote[p] - vote[m]));
                    rot[pass] = (float)bi + di + 0.5f;
                    ocount++;
                    if (pass == 0) maxIndex = bi;
                } else if (pass == 0) break;
            }
            k.nOri = 0;
            if (ocount > 0) {
                float fr1 = rot[0] / 36.0f; if (fr1 < 0) fr1 += 1.0f;
                const unsigned short us1 = (unsigned short)f2i(floorf(fr1 * 65535.0f));
                unsigned short us2 = 65535;
                if (ocount > 1) { float fr2 = rot[1] / 36.0f; if (fr2 < 0) fr2 += 1.0f; us2 = (unsigned short)f2i(floorf(fr2 * 65535.0f)); }
                const float fac = (float)(2.0 * 3.14159265358979323846 / 65535.0);
                // ReshapeFeatureList_Kernel :1994-2026
                if (us1 != 65535) {
                    k.ori[k.nOri++] = fac * (float)us1;
                    if (us2 != 65535 && us2 != us1) k.ori[k.nOri++] = fac * (float)us2;
                }
            }
        }
    }
    // ---- ReshapeFeatureList (scale gate) + LimitFeatureCount(1)
    struct Feat { float x, y, s, o; int li; };
    std::vector<Feat> feats[NUM_OCT * DOG_LEVELS];
    int finalNum[NUM_OCT * DOG_LEVELS];
    for (int li = 0; li < NUM_OCT * DOG_LEVELS; ++li) {
        finalNum[li] = 0;
        if (levelNum[li] == 0) continue;
        const int o = li / DOG_LEVELS, j = li % DOG_LEVELS;
        const float keyLocScale = (float)(1 << o);
        const float sigma = P.sigma0 * powf(2.0f, (float)j / (float)DOG_LEVELS);
        const int fmaxRaw = (int)(oct[o].w * oct[o].h * 0.005f);
        const int fmax = fmaxRaw > 4096 ? 4096 : (fmaxRaw < 32 ? 32 : fmaxRaw);
        for (const RawKey& k : raw[li]) {
            if (!(sigma * keyLocScale >= minKeyScale)) continue;
            for (int q = 0; q < k.nOri; ++q)
                if ((int)feats[li].size() < fmax) feats[li].push_back({k.x + 0.5f, k.y + 0.5f, sigma, k.ori[q], li});
        }
        finalNum[li] = (int)feats[li].size();
    }
    limit(finalNum);
    // ---- descriptors (ComputeDescriptor_Kernel :1178-1257, NormalizeDescriptor_Kernel :1339-1370),
    //      key list (CreateGlobalKeyPointList_Kernel :2049-2081), uchar (:2100-2107)
    int n = 0;
    for (int li = 0; li < NUM_OCT * DOG_LEVELS; ++li) {
        if (levelCounts) levelCounts[li] = finalNum[li];
        if (finalNum[li] == 0) continue;
        const int o = li / DOG_LEVELS, j = li % DOG_LEVELS;
        const Octave& oc = oct[o];
        const Level& L = oc.lev[j + 1];
        const int width = oc.w, height = oc.h;
        const float keyLocScale = (float)(1 << o), keyLocOffset = 0.5f;
        for (const Feat& f : feats[li]) {
            if (n >= maxFeatures) return -1;       // Bundler.cpp:97 "too many keypoints"
            float des[128];
            const float spt = fabsf(f.s * 3.0f);
            float s, c;
            bf_dm_sincos(f.o, &s, &c);
            const float anglef = f.o > PI_F ? f.o - (float)(2.0 * 3.14159265358979323846) : f.o;
            const float cspt = c * spt, sspt = s * spt, crspt = c / spt, srspt = s / spt;
            const float rpi = (float)(4.0 / 3.14159265358979323846);
            for (int cell = 0; cell < 16; ++cell) {
                const int ix = cell & 3, iy = cell >> 2;
                const float ox = ix - 1.5f, oy = iy - 1.5f;
                const float ptx = cspt * ox - sspt * oy + f.x, pty = cspt * oy + sspt * ox + f.y;
                const float bsz = fabsf(cspt) + fabsf(sspt);
                const float xmin = std::max(1.5f, floorf(ptx - bsz) + 0.5f), ymin = std::max(1.5f, floorf(pty - bsz) + 0.5f);
                const float xmax = std::min(width - 1.5f, floorf(ptx + bsz) + 0.5f), ymax = std::min(height - 1.5f, floorf(pty + bsz) + 0.5f);
                const unsigned xlen = f2u(roundf(xmax - xmin + 1)), ylen = f2u(roundf(ymax - ymin + 1));
                const unsigned size = xlen * ylen;
                static float part[8][64];
                for (int b = 0; b < 8; ++b) for (int l = 0; l < 64; ++l) part[b][l] = 0.0f;
                for (unsigned si = 0; si < size; ++si) {
                    const float x = (float)(si % xlen) + xmin, y = (float)(si / xlen) + ymin;
                    const float dx = x - ptx, dy = y - pty;
                    const float nx = crspt * dx + srspt * dy, ny = crspt * dy - srspt * dx;
                    const float nxn = fabsf(nx), nyn = fabsf(ny);
                    if (nxn < 1.0f && nyn < 1.0f) {
                        const size_t pi = (size_t)f2i(y) * width + (size_t)f2i(x);
                        const float dnx = nx + ox, dny = ny + oy;
                        const float ww = bf_dm_exp(-0.125f * (dnx * dnx + dny * dny));
                        const float wx = 1.0f - nxn, wy = 1.0f - nyn;
                        const float weight = ww * wx * wy * L.mag[pi];
                        float theta = (anglef - L.ang[pi]) * rpi;
                        if (theta < 0) theta += 8.0f;
                        const float fo = floorf(theta);
                        const int fidx = f2i(fo);
                        const float w1 = fo + 1.0f - theta, w2 = theta - fo;
                        part[fidx & 7][si & 63] += w1 * weight;
                        part[(fidx + 1) & 7][si & 63] += w2 * weight;
                    }
                }
                for (int b = 0; b < 8; ++b) des[cell * 8 + b] = butterfly(part[b]);
            }
            // normalise: 32 lanes x 4 values, xor butterfly over 32 lanes
            auto norm32 = [&](const float* v) {
                float lane[32];
                for (int t = 0; t < 32; ++t) lane[t] = v[4 * t] * v[4 * t] + v[4 * t + 1] * v[4 * t + 1] + v[4 * t + 2] * v[4 * t + 2] + v[4 * t + 3] * v[4 * t + 3];
                for (int o2 = 16; o2 > 0; o2 >>= 1) { float t2[32]; for (int t = 0; t < 32; ++t) t2[t] = lane[t] + lane[t ^ o2]; memcpy(lane, t2, sizeof t2); }
                return 1.0f / sqrtf(lane[0]);
            };
            const float n1 = norm32(des);
            for (int k = 0; k < 128; ++k) des[k] = std::min(0.2f, des[k] * n1);
            const float n2 = norm32(des);
            for (int k = 0; k < 128; ++k) des[k] *= n2;
            for (int k = 0; k < 128; ++k) descs[(size_t)n * 128 + k] = (uint8_t)f2i(512 * des[k] + 0.5f);
            const float posX = keyLocScale * (f.x - 0.5f) + keyLocOffset, posY = keyLocScale * (f.y - 0.5f) + keyLocOffset;
            const float depthX = posX * (float)(depthW - 1) / (float)(W - 1), depthY = posY * (float)(depthH - 1) / (float)(H - 1);
            const int ipx = f2i(roundf(depthX)), ipy = f2i(roundf(depthY));
            keys[4 * n + 0] = posX; keys[4 * n + 1] = posY; keys[4 * n + 2] = keyLocScale * f.s;
            keys[4 * n + 3] = depth[(size_t)ipy * depthW + ipx];
            ++n;
        }
    }
    return n;
}

// test hook: raw key lists after DetectKeypoints, before any limit: counts[12] and (col, row) pairs per slot (capacity keys per slot)
void or_sift_detect(const float* intensity, const float* depth, int W, int H, int depthW, int depthH, float depthMin, float depthMax, int* counts, int* xy, int capacity) {
    static SiftParams P = makeParams();
    std::vector<Octave> oct(NUM_OCT);
    std::vector<RawKey> raw[NUM_OCT * DOG_LEVELS];
    int levelNum[NUM_OCT * DOG_LEVELS];
    buildAndDetect(P, intensity, depth, W, H, depthW, depthH, depthMin, depthMax, oct, raw, levelNum);
    for (int li = 0; li < NUM_OCT * DOG_LEVELS; ++li) {
        counts[li] = levelNum[li];
        for (int k = 0; k < levelNum[li] && k < capacity; ++k) { xy[((size_t)li * capacity + k) * 2] = raw[li][k].x; xy[((size_t)li * capacity + k) * 2 + 1] = raw[li][k].y; }
    }
}

// test hook: one gaussian level of the pyramid (octave o, array index a in 0..5)
void or_sift_pyramid_level(const float* intensity, int W, int H, int o, int a, float* out) {
    static SiftParams P = makeParams();
    std::vector<float> cur;
    int w = W, h = H;
    std::vector<float> lev[NLEV];
    for (int oo = 0; oo <= o; ++oo) {
        if (oo == 0) { std::vector<float> in(intensity, intensity + (size_t)W * H); blur(in, lev[0], w, h, P.taps[0]); }
        else {
            std::vector<float> src = lev[3];
            const int pw = w; w >>= 1; h >>= 1;
            lev[0].assign((size_t)w * h, 0.0f);
            for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) lev[0][(size_t)y * w + x] = src[(size_t)(y << 1) * pw + std::min(x << 1, pw - 1)];
        }
        for (int j = 0; j <= LEVEL_MAX; ++j) blur(lev[j], lev[j + 1], w, h, P.taps[j + 1]);
    }
    memcpy(out, lev[a].data(), sizeof(float) * (size_t)w * h);
}

}  // extern "C"
