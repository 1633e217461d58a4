// Colour codecs for recorded sequences (.sens): baseline JPEG and PNG to RGB8, and a baseline JPEG encoder for recording; host only.
//
// ml::SensorData (mLib, not in the reference tree) decodes compressed colour frames with stb_image; this file is the
// library's built-in replacement so that a C++ host needs no image library to play a BundleFusion / ScanNet recording:
//   * JPEG: baseline / extended sequential DCT, 8 bit, Huffman, 1 or 3 components, sampling factors 1 or 2, restart intervals.
//     Inverse DCT is the accurate integer transform of the IJG code (the "islow" algorithm: Loeffler-Ligtenberg-Moschytz, 13-bit
//     constants), chroma is interpolated with the IJG triangle filter ("fancy upsampling"), YCbCr -> RGB in 16-bit fixed point -
//     i.e. the arithmetic of the decoder most files were checked with; stb_image differs from it by a few LSB at most.
//     Progressive and arithmetic-coded files are rejected.
//   * PNG: 8-bit grey / RGB / palette / with alpha (alpha dropped), 1/2/4-bit grey and palette, non-interlaced, all five filters.
// Lossy decoders are not bit-pinned by the reference; tests/test_sensordata_cpu.py compares against Pillow (libjpeg-turbo).
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/bf_sensordata.h"
#include "bf_internal.h"

using namespace bf;

namespace {

// ================================================================================================ JPEG
struct Huff {
    uint8_t bits[17];          // number of codes of each length 1..16
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
    }
};

struct Comp { int id, h, v, tq, td, ta, pred; int bw, bh; std::vector<uint8_t> plane; int pw, ph; };   // plane: padded to whole MCUs

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int cnt = 0; bool hitMarker = false;
    void fill() {
        while (cnt <= 24) {
            int b = 0;
            if (!hitMarker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;                 // stuffed zero
                    else { hitMarker = true; b = 0; }                         // a marker: feed zeros, do not consume it
                } else ++p;
            }
            acc |= (uint32_t)b << (24 - cnt);
            cnt += 8;
        }
    }
    int bit() { if (cnt == 0) fill(); const int b = (int)(acc >> 31); acc <<= 1; --cnt; return b; }
    int bits(int n) { if (n == 0) return 0; if (cnt < n) fill(); const int v = (int)(acc >> (32 - n)); acc <<= n; cnt -= n; return v; }
    void reset() { acc = 0; cnt = 0; hitMarker = false; }
};

int decodeSymbol(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

const int ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int descale(int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); }

// IJG jidctint ("islow"): two passes of the LL&M 8-point IDCT, CONST_BITS = 13, PASS1_BITS = 2; input already dequantised
void idctIslow(const int* in, uint8_t* out, int stride) {
    const int CB = 13, P1 = 2;
    const int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                  F2053 = 16819, F2562 = 20995, F3072 = 25172;
    int ws[64];
    for (int c = 0; c < 8; ++c) {
        const int* ip = in + c;
        if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
            const int dc = ip[0] * (1 << P1);
            for (int r = 0; r < 8; ++r) ws[r * 8 + c] = dc;
            continue;
        }
        int64_t z2 = ip[16], z3 = ip[48];
        int64_t z1 = (z2 + z3) * F0541;
        int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = ip[0]; z3 = ip[32];
        int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = ip[56]; tmp1 = ip[40]; tmp2 = ip[24]; tmp3 = ip[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
        ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
        ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
        ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const int* w = ws + r * 8;
        uint8_t* o = out + (size_t)r * stride;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F0541;
        int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CB + P1 + 3;
        o[0] = clamp8(descale(tmp10 + tmp3, S) + 128); o[7] = clamp8(descale(tmp10 - tmp3, S) + 128);
        o[1] = clamp8(descale(tmp11 + tmp2, S) + 128); o[6] = clamp8(descale(tmp11 - tmp2, S) + 128);
        o[2] = clamp8(descale(tmp12 + tmp1, S) + 128); o[5] = clamp8(descale(tmp12 - tmp1, S) + 128);
        o[3] = clamp8(descale(tmp13 + tmp0, S) + 128); o[4] = clamp8(descale(tmp13 - tmp0, S) + 128);
    }
}

// IJG "fancy" (triangle) up-sampling of one chroma plane to full resolution.  sw x sh: the down-sampled size that covers the image.
void upsampleH2(const uint8_t* in, int sw, uint8_t* out) {             // one row, 2:1 horizontally (h2v1_fancy_upsample)
    if (sw == 1) { out[0] = out[1] = in[0]; return; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < sw - 1; ++i) {
        out[2 * i] = (uint8_t)((in[i] * 3 + in[i - 1] + 1) >> 2);
        out[2 * i + 1] = (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
    }
    out[2 * (sw - 1)] = (uint8_t)((in[sw - 1] * 3 + in[sw - 2] + 1) >> 2);
    out[2 * (sw - 1) + 1] = in[sw - 1];
}

void upsampleH2V2Row(const uint8_t* near, const uint8_t* far, int sw, uint8_t* out) {      // h2v2_fancy_upsample: 3/4 near + 1/4 far vertically, then the triangle filter
    std::vector<int> col(sw);
    for (int i = 0; i < sw; ++i) col[i] = near[i] * 3 + far[i];
    if (sw == 1) { out[0] = out[1] = (uint8_t)((col[0] * 4 + 8) >> 4); return; }
    out[0] = (uint8_t)((col[0] * 4 + 8) >> 4);
    out[1] = (uint8_t)((col[0] * 3 + col[1] + 7) >> 4);
    for (int i = 1; i < sw - 1; ++i) {
        out[2 * i] = (uint8_t)((col[i] * 3 + col[i - 1] + 8) >> 4);
        out[2 * i + 1] = (uint8_t)((col[i] * 3 + col[i + 1] + 7) >> 4);
    }
    out[2 * (sw - 1)] = (uint8_t)((col[sw - 1] * 3 + col[sw - 2] + 8) >> 4);
    out[2 * (sw - 1) + 1] = (uint8_t)((col[sw - 1] * 4 + 7) >> 4);
}

int decodeJpeg(const uint8_t* data, size_t size, uint32_t width, uint32_t height, uint8_t* rgb) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { set_error("jpeg: no SOI marker"); return BF_ERR_INVALID_ARG; }
    uint16_t qt[4][64]; bool qtPresent[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0;
    size_t pos = 2;
    bool sawSOF = false;
    auto u16 = [&](size_t p) { return (int)data[p] << 8 | data[p + 1]; };
    while (pos + 4 <= size) {
        if (data[pos] != 0xFF) { ++pos; continue; }
        const int m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > size) break;
        const int len = u16(pos);
        if (len < 2 || pos + len > size) { set_error("jpeg: truncated segment"); return BF_ERR_INVALID_ARG; }
        const size_t seg = pos + 2, segEnd = pos + len;
        if (m == 0xDB) {                                                        // DQT
            size_t p = seg;
            while (p < segEnd) {
                const int pq = data[p] >> 4, tq = data[p] & 15; ++p;
                if (tq > 3 || p + (pq ? 128 : 64) > segEnd) { set_error("jpeg: bad DQT"); return BF_ERR_INVALID_ARG; }
                for (int i = 0; i < 64; ++i) { qt[tq][ZIGZAG[i]] = (uint16_t)(pq ? u16(p) : data[p]); p += pq ? 2 : 1; }
                qtPresent[tq] = true;
            }
        } else if (m == 0xC4) {                                                 // DHT
            size_t p = seg;
            while (p < segEnd) {
                const int tc = data[p] >> 4, th = data[p] & 15; ++p;
                if (tc > 1 || th > 3 || p + 16 > segEnd) { set_error("jpeg: bad DHT"); return BF_ERR_INVALID_ARG; }
                Huff& h = tc ? ac[th] : dc[th];
                int n = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = data[p++]; n += h.bits[l]; }
                if (n > 256 || p + n > segEnd) { set_error("jpeg: bad DHT"); return BF_ERR_INVALID_ARG; }
                memcpy(h.vals, data + p, n); p += n;
                h.build(); h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {                                     // SOF0 / SOF1
            if (seg + 6 > segEnd) { set_error("jpeg: short SOF segment"); return BF_ERR_INVALID_ARG; }      // precision, height, width, #components
            if (data[seg] != 8) { set_error("jpeg: only 8-bit samples are supported"); return BF_ERR_INVALID_ARG; }
            H = u16(seg + 1); W = u16(seg + 3);
            const int nc = data[seg + 5];
            if ((nc != 1 && nc != 3) || seg + 6 + 3 * nc > segEnd) { set_error("jpeg: %d components are not supported", nc); return BF_ERR_INVALID_ARG; }
            comps.resize(nc);
            for (int i = 0; i < nc; ++i) {
                Comp& c = comps[i];
                c.id = data[seg + 6 + 3 * i]; c.h = data[seg + 7 + 3 * i] >> 4; c.v = data[seg + 7 + 3 * i] & 15; c.tq = data[seg + 8 + 3 * i];
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) { set_error("jpeg: sampling factor %dx%d is not supported", c.h, c.v); return BF_ERR_INVALID_ARG; }
                hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
            }
            sawSOF = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            set_error("jpeg: progressive / lossless / arithmetic-coded files are not supported (SOF%d)", m - 0xC0);
            return BF_ERR_INVALID_ARG;
        } else if (m == 0xDD) {
            if (seg + 2 > segEnd) { set_error("jpeg: short DRI segment"); return BF_ERR_INVALID_ARG; }
            restart = u16(seg);
        } else if (m == 0xDA) {                                                 // SOS: the (single) scan follows
            if (!sawSOF) { set_error("jpeg: SOS before SOF"); return BF_ERR_INVALID_ARG; }
            if ((uint32_t)W != width || (uint32_t)H != height) { set_error("jpeg: image is %dx%d, expected %ux%u", W, H, width, height); return BF_ERR_INVALID_ARG; }
            if (seg + 1 > segEnd) { set_error("jpeg: short SOS segment"); return BF_ERR_INVALID_ARG; }
            const int ns = data[seg];
            if (ns != (int)comps.size()) { set_error("jpeg: non-interleaved scans are not supported"); return BF_ERR_INVALID_ARG; }
            if (seg + 1 + 2 * (size_t)ns + 3 > segEnd) { set_error("jpeg: short SOS segment"); return BF_ERR_INVALID_ARG; }      // component selectors + Ss, Se, Ah/Al
            for (int i = 0; i < ns; ++i) {
                const int cid = data[seg + 1 + 2 * i], t = data[seg + 2 + 2 * i];
                bool found = false;
                for (Comp& c : comps) if (c.id == cid) { c.td = t >> 4; c.ta = t & 15; found = true; }
                if (!found) { set_error("jpeg: scan names an unknown component"); return BF_ERR_INVALID_ARG; }
            }
            const int mcuW = 8 * hmax, mcuH = 8 * vmax, mcusX = (W + mcuW - 1) / mcuW, mcusY = (H + mcuH - 1) / mcuH;
            for (Comp& c : comps) {
                if (c.td > 3 || c.ta > 3 || !dc[c.td].present || !ac[c.ta].present || !qtPresent[c.tq]) { set_error("jpeg: missing Huffman or quantisation table"); return BF_ERR_INVALID_ARG; }
                c.pw = mcusX * c.h * 8; c.ph = mcusY * c.v * 8;
                c.plane.assign((size_t)c.pw * c.ph, 0);
                c.pred = 0;
            }
            BitReader br; br.p = data + segEnd; br.end = data + size;
            int untilRestart = restart;
            int coef[64];
            for (int my = 0; my < mcusY; ++my)
                for (int mx = 0; mx < mcusX; ++mx) {
                    if (restart && untilRestart == 0) {                          // RSTn: byte-align, skip the marker, reset predictors
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 1 < br.end) br.p += 2;
                        for (Comp& c : comps) c.pred = 0;
                        untilRestart = restart;
                    }
                    for (Comp& c : comps)
                        for (int by = 0; by < c.v; ++by)
                            for (int bx = 0; bx < c.h; ++bx) {
                                memset(coef, 0, sizeof coef);
                                const int t = decodeSymbol(br, dc[c.td]);
                                if (t < 0 || t > 11) { set_error("jpeg: corrupt DC code"); return BF_ERR_INVALID_ARG; }
                                const int diff = t ? extend(br.bits(t), t) : 0;
                                c.pred += diff;
                                coef[0] = c.pred * qt[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decodeSymbol(br, ac[c.ta]);
                                    if (rs < 0) { set_error("jpeg: corrupt AC code"); return BF_ERR_INVALID_ARG; }
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) { set_error("jpeg: corrupt AC run"); return BF_ERR_INVALID_ARG; }
                                    coef[ZIGZAG[k]] = extend(br.bits(s), s) * qt[c.tq][ZIGZAG[k]];
                                    ++k;
                                }
                                idctIslow(coef, c.plane.data() + (size_t)((my * c.v + by) * 8) * c.pw + (size_t)(mx * c.h + bx) * 8, c.pw);
                            }
                    if (restart) --untilRestart;
                }
            // ---- up-sample chroma and convert
            if (comps.size() == 1) {
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) { const uint8_t v = comps[0].plane[(size_t)y * comps[0].pw + x]; uint8_t* o = rgb + ((size_t)y * W + x) * 3; o[0] = o[1] = o[2] = v; }
                return BF_OK;
            }
            if (comps[0].h != hmax || comps[0].v != vmax) { set_error("jpeg: sub-sampled luma is not supported"); return BF_ERR_INVALID_ARG; }
            std::vector<uint8_t> full[2];
            for (int ci = 1; ci < 3; ++ci) {
                Comp& c = comps[ci];
                const int fh = hmax / c.h, fv = vmax / c.v;
                const int sw = (W * c.h + hmax - 1) / hmax, sh = (H * c.v + vmax - 1) / vmax;       // down-sampled size covering the image
                std::vector<uint8_t>& f = full[ci - 1];
                f.assign((size_t)(sw * fh + 2) * (sh * fv + 2), 0);
                const int fw = sw * fh;
                for (int y = 0; y < sh; ++y) {
                    const uint8_t* row = c.plane.data() + (size_t)y * c.pw;
                    if (fh == 1 && fv == 1) memcpy(f.data() + (size_t)y * fw, row, sw);
                    else if (fh == 2 && fv == 1) upsampleH2(row, sw, f.data() + (size_t)y * fw);
                    else if (fh == 2 && fv == 2) {
                        const uint8_t* up = c.plane.data() + (size_t)(y > 0 ? y - 1 : 0) * c.pw;
                        const uint8_t* dn = c.plane.data() + (size_t)(y < sh - 1 ? y + 1 : sh - 1) * c.pw;
                        upsampleH2V2Row(row, up, sw, f.data() + (size_t)(2 * y) * fw);
                        upsampleH2V2Row(row, dn, sw, f.data() + (size_t)(2 * y + 1) * fw);
                    } else {                                                    // 1:2 vertically only (rare): replicate rows
                        memcpy(f.data() + (size_t)(2 * y) * fw, row, sw);
                        memcpy(f.data() + (size_t)(2 * y + 1) * fw, row, sw);
                    }
                }
                c.bw = fw;
            }
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int Y = comps[0].plane[(size_t)y * comps[0].pw + x];
                    const int cb = full[0][(size_t)y * comps[1].bw + x] - 128, cr = full[1][(size_t)y * comps[2].bw + x] - 128;
                    uint8_t* o = rgb + ((size_t)y * W + x) * 3;                  // jdcolor.c build_ycc_rgb_table: 16-bit fixed point
                    o[0] = clamp8(Y + ((91881 * cr + 32768) >> 16));
                    o[1] = clamp8(Y + ((-22554 * cb - 46802 * cr + 32768) >> 16));
                    o[2] = clamp8(Y + ((116130 * cb + 32768) >> 16));
                }
            return BF_OK;
        }
        pos = segEnd;
    }
    set_error("jpeg: no scan found");
    return BF_ERR_INVALID_ARG;
}

// ================================================================================================ PNG
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int decodePng(const uint8_t* data, size_t size, uint32_t width, uint32_t height, uint8_t* rgb) {
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || memcmp(data, SIG, 8) != 0) { set_error("png: bad signature"); return BF_ERR_INVALID_ARG; }
    size_t pos = 8;
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    auto u32 = [&](size_t p) { return (uint32_t)data[p] << 24 | (uint32_t)data[p + 1] << 16 | (uint32_t)data[p + 2] << 8 | data[p + 3]; };
    bool end = false;
    while (!end && pos + 12 <= size) {
        const uint32_t len = u32(pos);
        if (pos + 12 + (size_t)len > size) { set_error("png: truncated chunk"); return BF_ERR_INVALID_ARG; }
        const uint8_t* type = data + pos + 4; const uint8_t* body = data + pos + 8;
        if (!memcmp(type, "IHDR", 4) && len >= 13) { W = u32(pos + 8); H = u32(pos + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (W != width || H != height) { set_error("png: image is %ux%u, expected %ux%u", W, H, width, height); return BF_ERR_INVALID_ARG; }
    if (interlace != 0) { set_error("png: interlaced images are not supported"); return BF_ERR_INVALID_ARG; }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (ctype == 3 && plte.size() < 3)) { set_error("png: colour type %d is not supported", ctype); return BF_ERR_INVALID_ARG; }
    if (!(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) { set_error("png: bit depth %d is not supported for colour type %d", depth, ctype); return BF_ERR_INVALID_ARG; }
    const size_t stride = ((size_t)W * ch * depth + 7) / 8;                      // bytes per scan line
    const size_t bpp = std::max<size_t>(1, (size_t)ch * depth / 8);              // filter distance in bytes
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) { set_error("png: the image data does not inflate to %zu bytes", raw.size()); return BF_ERR_INVALID_ARG; }
    std::vector<uint8_t> prev(stride, 0), cur(stride), px((size_t)W * ch);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t* line = raw.data() + (stride + 1) * y;
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
            int v;
            switch (ft) {
                case 0: v = x; break;
                case 1: v = x + a; break;
                case 2: v = x + b; break;
                case 3: v = x + ((a + b) >> 1); break;
                case 4: v = x + paeth(a, b, c); break;
                default: set_error("png: unknown filter type %d", ft); return BF_ERR_INVALID_ARG;
            }
            cur[i] = (uint8_t)v;
        }
        if (depth == 8) memcpy(px.data(), cur.data(), px.size());
        else {                                                                   // 1, 2 or 4 bits per sample (grey or palette index), MSB first
            const int mask = (1 << depth) - 1, scale = ctype == 0 ? 255 / mask : 1;
            for (uint32_t x = 0; x < W; ++x) {
                const size_t bit = (size_t)x * depth;
                px[x] = (uint8_t)(((cur[bit >> 3] >> (8 - depth - (bit & 7))) & mask) * scale);
            }
        }
        uint8_t* o = rgb + (size_t)y * W * 3;
        for (uint32_t x = 0; x < W; ++x) {
            const uint8_t* s = px.data() + (size_t)x * ch;
            if (ctype == 2 || ctype == 6) { o[3 * x] = s[0]; o[3 * x + 1] = s[1]; o[3 * x + 2] = s[2]; }
            else if (ctype == 3) { const size_t k = (size_t)s[0] * 3; if (k + 2 < plte.size()) { o[3 * x] = plte[k]; o[3 * x + 1] = plte[k + 1]; o[3 * x + 2] = plte[k + 2]; } else o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = 0; }
            else { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = s[0]; }
        }
        prev.swap(cur);
    }
    return BF_OK;
}

// ================================================================================================ JPEG encoder (recording)
// Baseline, 4:4:4, quantisation tables of Annex K scaled by `quality` like the IJG code, and Huffman tables optimised for the
// image (two passes; code lengths limited to 16 bits by the procedure of Annex K.2).  ml::SensorData records colour with
// stb_image_write's encoder; an encoder's output is not pinned by anything - any baseline decoder reads this one's.
const uint8_t STD_LUMA_Q[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t STD_CHROMA_Q[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

struct HuffEnc { uint8_t bits[17]; uint8_t vals[256]; int nvals; uint16_t code[256]; uint8_t len[256]; };

// optimal code lengths from symbol counts (Annex K.2 / jpeg_gen_optimal_table): a reserved pseudo-symbol keeps the all-ones code free
void buildOptimal(const long* freqIn, HuffEnc& h) {
    long freq[257]; int codesize[257], others[257];
    for (int i = 0; i < 256; ++i) freq[i] = freqIn[i];
    freq[256] = 1;
    for (int i = 0; i < 257; ++i) { codesize[i] = 0; others[i] = -1; }
    for (;;) {
        int c1 = -1, c2 = -1; long v = 1000000000L;
        for (int i = 0; i <= 256; ++i) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; ++i) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2]; freq[c2] = 0;
        codesize[c1]++; while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++; while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    int bits[33];
    for (int i = 0; i < 33; ++i) bits[i] = 0;
    for (int i = 0; i <= 256; ++i) if (codesize[i]) bits[codesize[i] > 32 ? 32 : codesize[i]]++;
    for (int i = 32; i > 16; --i)
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) --j;
            bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
    int i = 16;
    while (bits[i] == 0) --i;
    bits[i]--;                                                       // remove the pseudo-symbol's code
    h.bits[0] = 0;
    for (int l = 1; l <= 16; ++l) h.bits[l] = (uint8_t)bits[l];
    h.nvals = 0;
    for (int l = 1; l <= 32; ++l) for (int sym = 0; sym < 256; ++sym) if (codesize[sym] == l) h.vals[h.nvals++] = (uint8_t)sym;
    int code = 0, k = 0;
    for (int sym = 0; sym < 256; ++sym) h.len[sym] = 0;
    for (int l = 1; l <= 16; ++l) { for (int n = 0; n < h.bits[l]; ++n, ++k) { h.code[h.vals[k]] = (uint16_t)code++; h.len[h.vals[k]] = (uint8_t)l; } code <<= 1; }
}

struct BitWriter {
    std::vector<uint8_t>& out; uint32_t acc = 0; int cnt = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(uint32_t code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1)); cnt += len;
        while (cnt >= 8) { const uint8_t b = (uint8_t)(acc >> (cnt - 8)); out.push_back(b); if (b == 0xFF) out.push_back(0); cnt -= 8; }
    }
    void flush() { if (cnt) put(0x7F, 8 - cnt); }
};

inline int bitSize(int v) { v = v < 0 ? -v : v; int n = 0; while (v) { ++n; v >>= 1; } return n; }

void fdct8x8(float* d) {                                              // separable DCT-II, orthonormal scaling folded into the output (direct form: 8x8x8x2)
    static float C[8][8]; static bool init = false;
    if (!init) { for (int k = 0; k < 8; ++k) for (int n = 0; n < 8; ++n) C[k][n] = (k == 0 ? 0.35355339059327373f : 0.5f) * (float)std::cos((2 * n + 1) * k * 3.14159265358979323846 / 16.0); init = true; }
    float t[64];
    for (int r = 0; r < 8; ++r) for (int k = 0; k < 8; ++k) { float s = 0; for (int n = 0; n < 8; ++n) s += C[k][n] * d[r * 8 + n]; t[r * 8 + k] = s; }
    for (int c = 0; c < 8; ++c) for (int k = 0; k < 8; ++k) { float s = 0; for (int n = 0; n < 8; ++n) s += C[k][n] * t[n * 8 + c]; d[k * 8 + c] = s; }
}

int encodeJpeg(const uint8_t* rgb, uint32_t W, uint32_t H, int quality, std::vector<uint8_t>& out) {
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t q[2][64];
    for (int i = 0; i < 64; ++i) {
        int a = ((int)STD_LUMA_Q[i] * scale + 50) / 100, b = ((int)STD_CHROMA_Q[i] * scale + 50) / 100;
        q[0][i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a)); q[1][i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    const uint32_t bx = (W + 7) / 8, by = (H + 7) / 8;
    std::vector<int16_t> coef((size_t)bx * by * 3 * 64);              // quantised coefficients in zigzag order, per block: Y, Cb, Cr
    float blk[3][64];
    for (uint32_t y0 = 0; y0 < by; ++y0)
        for (uint32_t x0 = 0; x0 < bx; ++x0) {
            for (int yy = 0; yy < 8; ++yy)
                for (int xx = 0; xx < 8; ++xx) {
                    const uint32_t x = std::min(x0 * 8 + xx, W - 1), y = std::min(y0 * 8 + yy, H - 1);      // edge replication
                    const uint8_t* p = rgb + ((size_t)y * W + x) * 3;
                    const float r = p[0], g = p[1], b = p[2];
                    blk[0][yy * 8 + xx] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
                    blk[1][yy * 8 + xx] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                    blk[2][yy * 8 + xx] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                }
            for (int c = 0; c < 3; ++c) {
                fdct8x8(blk[c]);
                int16_t* o = coef.data() + (((size_t)y0 * bx + x0) * 3 + c) * 64;
                const uint8_t* qt = q[c ? 1 : 0];
                for (int i = 0; i < 64; ++i) { const float v = blk[c][ZIGZAG[i]] / (float)qt[ZIGZAG[i]]; o[i] = (int16_t)(v < 0 ? -(int)(-v + 0.5f) : (int)(v + 0.5f)); }
            }
        }
    // pass 1: symbol statistics; pass 2: emit
    HuffEnc dcT[2], acT[2];
    std::vector<uint8_t> scan;
    for (int pass = 0; pass < 2; ++pass) {
        long fdc[2][256], fac[2][256];
        memset(fdc, 0, sizeof fdc); memset(fac, 0, sizeof fac);
        BitWriter bw(scan);
        int pred[3] = {0, 0, 0};
        for (size_t b = 0; b < (size_t)bx * by; ++b)
            for (int c = 0; c < 3; ++c) {
                const int16_t* o = coef.data() + (b * 3 + c) * 64;
                const int t = c ? 1 : 0;
                const int diff = o[0] - pred[c]; pred[c] = o[0];
                const int s = bitSize(diff);
                if (pass == 0) fdc[t][s]++; else { bw.put(dcT[t].code[s], dcT[t].len[s]); if (s) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s); }
                int run = 0;
                for (int k = 1; k < 64; ++k) {
                    const int v = o[k];
                    if (v == 0) { ++run; continue; }
                    while (run > 15) { if (pass == 0) fac[t][0xF0]++; else bw.put(acT[t].code[0xF0], acT[t].len[0xF0]); run -= 16; }
                    const int sz = bitSize(v), sym = (run << 4) | sz;
                    if (pass == 0) fac[t][sym]++; else { bw.put(acT[t].code[sym], acT[t].len[sym]); bw.put((uint32_t)(v < 0 ? v - 1 : v), sz); }
                    run = 0;
                }
                if (run) { if (pass == 0) fac[t][0]++; else bw.put(acT[t].code[0], acT[t].len[0]); }
            }
        if (pass == 0) { for (int t = 0; t < 2; ++t) { buildOptimal(fdc[t], dcT[t]); buildOptimal(fac[t], acT[t]); } }
        else bw.flush();
    }
    auto u16 = [&](int v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); };
    out.clear();
    out.push_back(0xFF); out.push_back(0xD8);
    static const uint8_t JFIF[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    out.insert(out.end(), JFIF, JFIF + sizeof JFIF);
    for (int t = 0; t < 2; ++t) { out.push_back(0xFF); out.push_back(0xDB); u16(67); out.push_back((uint8_t)t); for (int i = 0; i < 64; ++i) out.push_back(q[t][ZIGZAG[i]]); }
    out.push_back(0xFF); out.push_back(0xC0); u16(17); out.push_back(8); u16((int)H); u16((int)W); out.push_back(3);
    for (int c = 0; c < 3; ++c) { out.push_back((uint8_t)(c + 1)); out.push_back(0x11); out.push_back((uint8_t)(c ? 1 : 0)); }
    for (int t = 0; t < 2; ++t)
        for (int cls = 0; cls < 2; ++cls) {
            const HuffEnc& h = cls ? acT[t] : dcT[t];
            out.push_back(0xFF); out.push_back(0xC4); u16(2 + 1 + 16 + h.nvals); out.push_back((uint8_t)((cls << 4) | t));
            for (int l = 1; l <= 16; ++l) out.push_back(h.bits[l]);
            out.insert(out.end(), h.vals, h.vals + h.nvals);
        }
    out.push_back(0xFF); out.push_back(0xDA); u16(12); out.push_back(3);
    for (int c = 0; c < 3; ++c) { out.push_back((uint8_t)(c + 1)); out.push_back((uint8_t)(c ? 0x11 : 0x00)); }
    out.push_back(0); out.push_back(63); out.push_back(0);
    out.insert(out.end(), scan.begin(), scan.end());
    out.push_back(0xFF); out.push_back(0xD9);
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_decode_color_rgb(const uint8_t* data, uint64_t size, int32_t compressionType, uint32_t width, uint32_t height, uint8_t* rgbOut) try {
    BF_REQUIRE(data && rgbOut && width > 0 && height > 0, "null argument");
    if (compressionType == BF_SENS_COLOR_JPEG) return decodeJpeg(data, (size_t)size, width, height, rgbOut);
    if (compressionType == BF_SENS_COLOR_PNG) return decodePng(data, (size_t)size, width, height, rgbOut);
    if (compressionType == BF_SENS_COLOR_RAW) {
        BF_REQUIRE(size == (uint64_t)width * height * 3, "raw colour has the wrong size");
        memcpy(rgbOut, data, (size_t)size);
        return BF_OK;
    }
    set_error("colour compression type %d is not supported", compressionType);
    return BF_ERR_INVALID_ARG;
} catch (const std::exception& e) {                   // e.g. std::bad_alloc on an absurd image size: no exception leaves the C ABI
    set_error("colour decoder: %s", e.what());
    return BF_ERR_STATE;
}

// baseline JPEG of a width x height RGB8 image.  Call with out == NULL to get the size; the encoded stream is kept until the next call
// of this function on the same thread.
int bf_encode_jpeg_rgb(const uint8_t* rgb, uint32_t width, uint32_t height, int32_t quality, uint8_t* out, uint64_t capacity, uint64_t* size) try {
    BF_REQUIRE(rgb && size && width > 0 && height > 0 && width < 65536 && height < 65536, "bad argument");
    static thread_local std::vector<uint8_t> buf;
    static thread_local const uint8_t* last = nullptr;
    if (!out || last != rgb || buf.empty()) { const int rc = encodeJpeg(rgb, width, height, quality, buf); if (rc) return rc; last = rgb; }
    *size = buf.size();
    if (!out) return BF_OK;
    BF_REQUIRE(capacity >= buf.size(), "buffer too small");
    memcpy(out, buf.data(), buf.size());
    last = nullptr;
    return BF_OK;
} catch (const std::exception& e) {
    set_error("jpeg encoder: %s", e.what());
    return BF_ERR_STATE;
}

}  // extern "C"
