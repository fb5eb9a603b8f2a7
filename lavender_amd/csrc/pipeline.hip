// Input pipeline of the pretrain path on the MI355X (include/lavender_pipeline.h): TSV rows -> base64 -> JPEG entropy decode
// on host threads -> [PCIe: int16 DCT coefficients] -> dequantise + IDCT, chroma upsampling, colour conversion, antialiased
// resize, crop and normalisation on the GPU.
//
// The integer arithmetic of every stage follows the published algorithm of the library the reference calls, so that the
// frames equal the reference's bit for bit (dataset.py:177-186 cv2.imdecode / PIL = libjpeg(-turbo) defaults: JDCT_ISLOW,
// fancy upsampling; dataset.py:107-175 torchvision Resize on a PIL image = Pillow ImagingResample, 8 bits per channel):
//   * inverse DCT: Loeffler-Ligtenberg-Moschytz 8x8, 13-bit constants, 2 extra bits after the column pass, output
//     (x + 2^17) >> 18 + 128 through libjpeg's wrap-around range table;
//   * h2v2 "fancy" upsampling: 3:1 vertical then 3:1 horizontal triangle filter, roundings 8 / 7 alternating, edge rows and
//     columns replicated; h2v1: 3:1 horizontal, roundings 1 / 2;
//   * YCbCr -> RGB: 16-bit fixed point (1.40200, 1.77200, 0.71414, 0.34414), one rounding per product table;
//   * resize: per output pixel the normalised triangle weights over `support = max(scale, 1)` source pixels (double), turned
//     into 22-bit integers, horizontal pass to 8 bits, then vertical pass to 8 bits.
// Data layout in HBM (per batch): coefficient blocks [frame][component][block row][block col][64] int16 in natural order;
// component planes u8 padded to whole MCUs; RGBX frames (4 bytes per pixel, after the optional zero padding); the horizontal
// pass output holds only the rows and columns the crop window needs.  All kernels are HBM-bound byte / integer work: one
// thread per 8x8 block (IDCT) or per pixel, coalesced 4- / 16-byte accesses, frames as blockIdx.y.
#include "common.h"
#include "../../include/lavender_pipeline.h"

#include <fcntl.h>
#include <math.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

// =====================================================================================================================
// TSV
// =====================================================================================================================
namespace {
struct Tsv {
    int fd = -1;
    const char* base = nullptr;
    size_t size = 0;
    std::vector<long> idx;
};
}  // namespace

extern "C" void* lav_tsv_open(const char* tsv_path, const char* lineidx_path) {
    if (!tsv_path) { lav_set_error("lav_tsv_open: no path"); return nullptr; }
    Tsv* t = new Tsv;
    t->fd = open(tsv_path, O_RDONLY);
    struct stat st;
    if (t->fd < 0 || fstat(t->fd, &st) != 0) { lav_set_error("lav_tsv_open: cannot open %s", tsv_path); delete t; return nullptr; }
    t->size = (size_t)st.st_size;
    if (t->size) {
        void* p = mmap(nullptr, t->size, PROT_READ, MAP_PRIVATE, t->fd, 0);
        if (p == MAP_FAILED) { lav_set_error("lav_tsv_open: mmap of %s failed", tsv_path); close(t->fd); delete t; return nullptr; }
        t->base = (const char*)p;
    }
    if (lineidx_path) {
        FILE* f = fopen(lineidx_path, "r");
        if (!f) { lav_set_error("lav_tsv_open: cannot open %s", lineidx_path); lav_tsv_close(t); return nullptr; }
        long v;
        while (fscanf(f, "%ld", &v) == 1) {
            if (v < 0 || (size_t)v >= t->size) { lav_set_error("lav_tsv_open: offset %ld outside %s", v, tsv_path); fclose(f); lav_tsv_close(t); return nullptr; }
            t->idx.push_back(v);
        }
        fclose(f);
    } else {
        for (size_t p = 0; p < t->size;) {
            t->idx.push_back((long)p);
            const char* nl = (const char*)memchr(t->base + p, '\n', t->size - p);
            if (!nl) break;
            p = (size_t)(nl - t->base) + 1;
        }
    }
    return t;
}
extern "C" long lav_tsv_rows(void* h) { return h ? (long)((Tsv*)h)->idx.size() : -1; }
extern "C" long lav_tsv_row_offset(void* h, long row) {
    Tsv* t = (Tsv*)h;
    if (!t || row < 0 || row >= (long)t->idx.size()) { lav_set_error("lav_tsv_row_offset: row %ld out of range", row); return -1; }
    return t->idx[row];
}
extern "C" int lav_tsv_fields(void* h, long pos, int max_fields, const char** field, long* field_len) {
    Tsv* t = (Tsv*)h;
    LAV_REQUIRE(t && pos >= 0 && (size_t)pos < t->size && max_fields >= 0 && (max_fields == 0 || (field && field_len)), "lav_tsv_fields: bad arguments");
    const char* p = t->base + pos;
    const char* end = (const char*)memchr(p, '\n', t->size - pos);
    if (!end) end = t->base + t->size;
    int n = 0;
    while (true) {
        const char* tab = (const char*)memchr(p, '\t', end - p);
        const char* fe = tab ? tab : end;
        const char* a = p; const char* b = fe;
        while (a < b && (*a == ' ' || *a == '\r' || *a == '\n' || *a == '\t' || *a == '\v' || *a == '\f')) ++a;
        while (b > a && (b[-1] == ' ' || b[-1] == '\r' || b[-1] == '\n' || b[-1] == '\t' || b[-1] == '\v' || b[-1] == '\f')) --b;
        if (n < max_fields) { field[n] = a; field_len[n] = b - a; }
        ++n;
        if (!tab) break;
        p = tab + 1;
    }
    return n;
}
extern "C" void lav_tsv_close(void* h) {
    Tsv* t = (Tsv*)h;
    if (!t) return;
    if (t->base) munmap((void*)t->base, t->size);
    if (t->fd >= 0) close(t->fd);
    delete t;
}

// =====================================================================================================================
// base64 + JPEG entropy decoding (host)
// =====================================================================================================================
namespace {
struct B64Table {
    int8_t v[256];
    B64Table() {
        memset(v, -1, sizeof v);
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) v[(uint8_t)a[i]] = (int8_t)i;
    }
};
const B64Table B64;

// characters outside the alphabet are skipped (binascii.a2b_base64 semantics the reference relies on), '=' ends the data
long b64_decode(const char* s, long n, uint8_t* out, long cap) {
    uint32_t acc = 0; int bits = 0; long o = 0;
    for (long i = 0; i < n; ++i) {
        const uint8_t c = (uint8_t)s[i];
        if (c == '=') break;
        const int v = B64.v[c];
        if (v < 0) continue;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) {
            bits -= 8;
            if (o >= cap) return -1;
            out[o++] = (uint8_t)(acc >> bits);
        }
    }
    return o;
}

const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17], vals[256];
    uint16_t fast[512];          // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    int maxcode[18], valptr[17], mincode[17];
    void build() {
        int code = 0, k = 0;
        memset(fast, 0, sizeof fast);
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < bits[l]; ++i, ++k, ++code)
                if (l <= 9)
                    for (int f = 0; f < (1 << (9 - l)); ++f) fast[(code << (9 - l)) | f] = (uint16_t)((l << 8) | vals[k]);
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct JpegHeader {
    int w = 0, h = 0, ncomp = 0;
    int id[3], hs[3], vs[3], tq[3], td[3], ta[3];
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    int bw[3], bh[3];            // block grid per component (whole MCUs)
    uint16_t q[4][64];           // natural order
    bool qset[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart = 0;
    long scan = -1;              // offset of the entropy-coded data
    long coef_count() const { long n = 0; for (int c = 0; c < ncomp; ++c) n += (long)bw[c] * bh[c] * 64; return n; }
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Parses the markers up to and including the (single, interleaved) start of scan.  Returns nullptr or an error text.
const char* jpeg_parse(const uint8_t* d, long n, JpegHeader& H, bool tables) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return "not a JPEG (no SOI)";
    long p = 2;
    bool sof = false;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return "marker expected";
        while (p < n && d[p] == 0xFF) ++p;                      // fill bytes
        if (p >= n) break;
        const int m = d[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return "end of image before the scan";
        if (p + 2 > n) break;
        const int L = be16(d + p);
        if (L < 2 || p + L > n) return "truncated marker segment";
        const uint8_t* s = d + p + 2; const int sl = L - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return "only 8-bit samples are supported";
            H.h = be16(s + 1); H.w = be16(s + 3); H.ncomp = s[5];
            if (H.w <= 0 || H.h <= 0) return "empty frame";
            if (H.ncomp != 1 && H.ncomp != 3) return "only 1- and 3-component JPEGs are supported";
            if (sl < 6 + 3 * H.ncomp) return "truncated frame header";
            for (int c = 0; c < H.ncomp; ++c) {
                H.id[c] = s[6 + 3 * c]; H.hs[c] = s[7 + 3 * c] >> 4; H.vs[c] = s[7 + 3 * c] & 15; H.tq[c] = s[8 + 3 * c] & 3;
                if (H.hs[c] < 1 || H.vs[c] < 1) return "bad sampling factor";
            }
            if (H.ncomp == 1) { H.hs[0] = H.vs[0] = 1; }
            else {
                const bool ok = H.hs[1] == 1 && H.vs[1] == 1 && H.hs[2] == 1 && H.vs[2] == 1 &&
                                ((H.hs[0] == 1 && H.vs[0] == 1) || (H.hs[0] == 2 && H.vs[0] == 1) || (H.hs[0] == 2 && H.vs[0] == 2));
                if (!ok) return "chroma subsampling other than 4:4:4 / 4:2:2 / 4:2:0 is not supported";
            }
            H.hmax = H.hs[0]; H.vmax = H.vs[0];
            H.mcux = (H.w + 8 * H.hmax - 1) / (8 * H.hmax); H.mcuy = (H.h + 8 * H.vmax - 1) / (8 * H.vmax);
            for (int c = 0; c < H.ncomp; ++c) { H.bw[c] = H.mcux * H.hs[c]; H.bh[c] = H.mcuy * H.vs[c]; }
            sof = true;
            if (!tables) return nullptr;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return "progressive / lossless / arithmetic-coded JPEGs are not supported (baseline Huffman only)";
        } else if (m == 0xDB) {
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3 || o + 1 + 64 * (pq ? 2 : 1) > sl) return "bad quantisation table";
                for (int i = 0; i < 64; ++i) H.q[tq][ZIGZAG[i]] = (uint16_t)(pq ? be16(s + o + 1 + 2 * i) : s[o + 1 + i]);
                H.qset[tq] = true;
                o += 1 + 64 * (pq ? 2 : 1);
            }
        } else if (m == 0xC4) {
            int o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return "bad Huffman table id";
                Huff& T = tc ? H.ac[th] : H.dc[th];
                int cnt = 0; T.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { T.bits[i] = s[o + i]; cnt += s[o + i]; }
                if (cnt > 256 || o + 17 + cnt > sl) return "bad Huffman table";
                memcpy(T.vals, s + o + 17, cnt);
                T.present = true; T.build();
                o += 17 + cnt;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return "bad restart interval";
            H.restart = be16(s);
        } else if (m == 0xDA) {
            if (!sof) return "scan before the frame header";
            if (sl < 1 || s[0] != H.ncomp || sl < 1 + 2 * H.ncomp + 3) return "non-interleaved scans are not supported";
            for (int i = 0; i < H.ncomp; ++i) {
                int c = -1;
                for (int k = 0; k < H.ncomp; ++k) if (H.id[k] == s[1 + 2 * i]) c = k;
                if (c != i) return "scan component order differs from the frame header";
                H.td[c] = s[2 + 2 * i] >> 4; H.ta[c] = s[2 + 2 * i] & 15;
                if (H.td[c] > 3 || H.ta[c] > 3 || !H.dc[H.td[c]].present || !H.ac[H.ta[c]].present) return "scan refers to a missing Huffman table";
                if (!H.qset[H.tq[c]]) return "component refers to a missing quantisation table";
            }
            H.scan = p + L;
            return nullptr;
        }
        p += L;
    }
    return sof && !tables ? nullptr : "no scan found";
}

struct BitReader {
    const uint8_t* d; long n, p;
    uint64_t acc = 0; int cnt = 0;
    bool hit_marker = false;
    void fill() {
        while (cnt <= 56) {
            uint32_t b = 0;
            if (!hit_marker && p < n) {
                b = d[p];
                if (b == 0xFF) {
                    if (p + 1 < n && d[p + 1] == 0x00) p += 2;
                    else { hit_marker = true; b = 0; }
                } else ++p;
            }
            acc |= (uint64_t)b << (56 - cnt);
            cnt += 8;
        }
    }
    inline uint32_t peek(int k) { return (uint32_t)(acc >> (64 - k)); }
    inline void skip(int k) { acc <<= k; cnt -= k; }
    inline int receive_extend(int s) {                      // s in 1..15
        if (cnt < s) fill();
        const int v = (int)peek(s); skip(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    inline int decode(const Huff& T) {
        if (cnt < 16) fill();
        const uint16_t f = T.fast[peek(9)];
        if (f) { skip(f >> 8); return f & 255; }
        int code = (int)peek(16);
        for (int l = 10; l <= 16; ++l) {
            const int c = code >> (16 - l);
            if (T.maxcode[l] >= 0 && c <= T.maxcode[l] && c >= T.mincode[l]) { skip(l); return T.vals[T.valptr[l] + c - T.mincode[l]]; }
        }
        return -1;
    }
    // restart marker: drop the remaining bits of the current byte and the marker itself
    bool restart() {
        acc = 0; cnt = 0;
        if (hit_marker) {
            if (p + 1 < n && d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7) { p += 2; hit_marker = false; return true; }
            return false;
        }
        // the byte-aligned position is p minus the bytes still buffered: all were consumed above (acc cleared), but bytes read
        // ahead of the marker cannot exist (fill stops at a marker), so p points at the marker or at padding
        while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
        if (p + 1 >= n) return false;
        p += 2;
        return true;
    }
};

// coef: [component][block row][block col][64] int16, natural order, zero-initialised here
const char* jpeg_entropy_decode(const uint8_t* d, long n, const JpegHeader& H, int16_t* coef) {
    memset(coef, 0, sizeof(int16_t) * (size_t)H.coef_count());
    int16_t* base[3]; long off = 0;
    for (int c = 0; c < H.ncomp; ++c) { base[c] = coef + off; off += (long)H.bw[c] * H.bh[c] * 64; }
    BitReader br{d, n, H.scan};
    int pred[3] = {0, 0, 0};
    int togo = H.restart;
    for (int my = 0; my < H.mcuy; ++my)
        for (int mx = 0; mx < H.mcux; ++mx) {
            if (H.restart && togo == 0) {
                if (!br.restart()) return "restart marker missing";
                pred[0] = pred[1] = pred[2] = 0; togo = H.restart;
            }
            for (int c = 0; c < H.ncomp; ++c) {
                const Huff& DC = H.dc[H.td[c]]; const Huff& AC = H.ac[H.ta[c]];
                for (int by = 0; by < H.vs[c]; ++by)
                    for (int bx = 0; bx < H.hs[c]; ++bx) {
                        int16_t* blk = base[c] + ((long)(my * H.vs[c] + by) * H.bw[c] + (mx * H.hs[c] + bx)) * 64;
                        int s = br.decode(DC);
                        if (s < 0 || s > 15) return "corrupt entropy data (DC)";
                        if (s) pred[c] += br.receive_extend(s);
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = br.decode(AC);
                            if (rs < 0) return "corrupt entropy data (AC)";
                            const int r = rs >> 4; s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                k += 16;
                            } else {
                                k += r;
                                if (k > 63) return "corrupt entropy data (run past the block)";
                                blk[ZIGZAG[k]] = (int16_t)br.receive_extend(s);
                                ++k;
                            }
                        }
                    }
            }
            --togo;
        }
    return nullptr;
}
}  // namespace

extern "C" int lav_jpeg_peek(const char* b64, long b64_len, int* width, int* height) {
    LAV_REQUIRE(b64 && b64_len > 0 && width && height, "lav_jpeg_peek: bad arguments");
    // the frame header sits behind the application segments and tables: decode in growing prefixes
    std::vector<uint8_t> buf;
    for (long take = 2048; ; take *= 8) {
        const long nin = take < b64_len ? take : b64_len;
        buf.resize((size_t)nin);
        const long n = b64_decode(b64, nin, buf.data(), (long)buf.size());
        JpegHeader H;
        const char* err = n < 0 ? "base64 decode failed" : jpeg_parse(buf.data(), n, H, false);
        if (!err && H.w > 0) { *width = H.w; *height = H.h; return 0; }
        if (nin == b64_len) { lav_set_error("lav_jpeg_peek: %s", err ? err : "no frame header"); return LAV_E_ARG; }
    }
}

// =====================================================================================================================
// device stage
// =====================================================================================================================
namespace {
struct FrameDesc {
    int w, h, ncomp, sub;            // sub: 0 = 4:4:4 or grey, 1 = h2v1, 2 = h2v2
    int bw[3], bh[3];                // block grid per component
    long coef_off[3];                // int16 elements into the coefficient buffer
    long plane_off[3];               // bytes into the plane buffer (plane width = 8 * bw)
    int cw, ch;                      // real chroma samples per row / rows (ceil(w / 2), ceil(h / 2) as subsampled)
    int pad_l, pad_t, pw, ph;        // padded frame (pw x ph) and where the decoded frame sits in it
    long rgb_off;                    // pixels (4 bytes each) into the RGBX buffer
    int rw, rh;                      // resized size
    int crop_x, crop_y;
    int kx, ky;                      // coefficients per output pixel, horizontal / vertical
    long bx_off, kx_off, by_off, ky_off;   // ints into the table buffer: bounds (xmin, count) per output pixel of the crop; coefficients
    int row0, nrows;                 // source rows [row0, row0 + nrows) of the padded frame feed the vertical pass
    int need_h, need_v;
    long tmp_off;                    // pixels into the horizontal-pass buffer: nrows x out_w
    long out_off;                    // floats into the output tensor
    uint16_t q[3][64];
};

#define C_BITS 13
#define P1_BITS 2
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass of the "islow" inverse DCT; v[] in, o[] out before descaling
__device__ __forceinline__ void idct8(const int* v, int* o) {
    int z2 = v[2], z3 = v[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137);
    int tmp3 = z1 + z2 * 6270;
    z2 = v[0]; z3 = v[4];
    int tmp0 = (z2 + z3) << C_BITS;
    int tmp1 = (z2 - z3) << C_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3; o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1; o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}
// libjpeg's post-IDCT range table indexed with (x & 1023): 0..127 -> +128, 128..511 -> 255, 512..895 -> 0, 896.. -> x - 896
__device__ __forceinline__ int idct_range(int x) {
    const int y = x & 1023;
    return y < 128 ? y + 128 : (y < 512 ? 255 : (y < 896 ? 0 : y - 896));
}

__global__ __launch_bounds__(64) void jpeg_idct_kernel(const FrameDesc* desc, const int16_t* coef, uint8_t* planes) {
    const FrameDesc& f = desc[blockIdx.y];
    int b = blockIdx.x * 64 + threadIdx.x, c = 0;
    for (; c < f.ncomp; ++c) {
        const int nb = f.bw[c] * f.bh[c];
        if (b < nb) break;
        b -= nb;
    }
    if (c >= f.ncomp) return;
    const uint4* src = (const uint4*)(coef + f.coef_off[c] + (long)b * 64);
    int ws[64];
#pragma unroll
    for (int r = 0; r < 8; ++r) {                          // row r of the block: 8 coefficients = 16 bytes
        const uint4 u = src[r];
        const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ws[r * 8 + 2 * i] = (int)(int16_t)(w4[i] & 0xffff) * (int)f.q[c][r * 8 + 2 * i];
            ws[r * 8 + 2 * i + 1] = (int)(int16_t)(w4[i] >> 16) * (int)f.q[c][r * 8 + 2 * i + 1];
        }
    }
#pragma unroll
    for (int col = 0; col < 8; ++col) {                    // pass 1: columns
        int v[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ws[r * 8 + col];
        idct8(v, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r * 8 + col] = descale(o[r], C_BITS - P1_BITS);
    }
    const int pwid = f.bw[c] * 8;
    uint8_t* dst = planes + f.plane_off[c] + (long)(b / f.bw[c]) * 8 * pwid + (b % f.bw[c]) * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {                          // pass 2: rows
        int o[8];
        idct8(ws + r * 8, o);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo |= (uint32_t)idct_range(descale(o[i], C_BITS + P1_BITS + 3)) << (8 * i);
            hi |= (uint32_t)idct_range(descale(o[4 + i], C_BITS + P1_BITS + 3)) << (8 * i);
        }
        *(uint2*)(dst + (long)r * pwid) = make_uint2(lo, hi);
    }
}

__device__ __forceinline__ int clamp8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// "fancy" chroma sample at full-resolution position (x, y)
__device__ __forceinline__ int chroma_at(const uint8_t* p, int pwid, int cw, int ch, int sub, int x, int y) {
    if (sub == 0) return p[(long)y * pwid + x];
    const int i = x >> 1;
    if (sub == 1) {                                        // h2v1: 3/4 nearer + 1/4 further, roundings 1 (even) / 2 (odd)
        const uint8_t* r = p + (long)y * pwid;
        const int cur = r[i];
        if (x & 1) return i + 1 < cw ? (cur * 3 + r[i + 1] + 2) >> 2 : cur;
        return i > 0 ? (cur * 3 + r[i - 1] + 1) >> 2 : cur;
    }
    const int j = y >> 1;                                  // h2v2: vertical 3:1 first (16-bit column sums), then horizontal 3:1
    int jn = (y & 1) ? j + 1 : j - 1;
    jn = jn < 0 ? 0 : (jn >= ch ? ch - 1 : jn);
    const uint8_t* r0 = p + (long)j * pwid;
    const uint8_t* r1 = p + (long)jn * pwid;
    const int cur = r0[i] * 3 + r1[i];
    if (x & 1) return i + 1 < cw ? (cur * 3 + r0[i + 1] * 3 + r1[i + 1] + 7) >> 4 : (cur * 4 + 7) >> 4;
    return i > 0 ? (cur * 3 + r0[i - 1] * 3 + r1[i - 1] + 8) >> 4 : (cur * 4 + 8) >> 4;
}

// planes -> RGBX frame of the padded size (zero border: torchvision Pad fill = 0)
__global__ __launch_bounds__(256) void jpeg_color_kernel(const FrameDesc* desc, const uint8_t* planes, uint32_t* rgb) {
    const FrameDesc& f = desc[blockIdx.y];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)f.pw * f.ph) return;
    const int px = (int)(idx % f.pw), py = (int)(idx / f.pw);
    const int x = px - f.pad_l, y = py - f.pad_t;
    uint32_t v = 0;
    if (x >= 0 && x < f.w && y >= 0 && y < f.h) {
        const int Y = planes[f.plane_off[0] + (long)y * (f.bw[0] * 8) + x];
        if (f.ncomp == 1) v = (uint32_t)Y * 0x010101u;
        else {
            const int cb = chroma_at(planes + f.plane_off[1], f.bw[1] * 8, f.cw, f.ch, f.sub, x, y) - 128;
            const int cr = chroma_at(planes + f.plane_off[2], f.bw[2] * 8, f.cw, f.ch, f.sub, x, y) - 128;
            const int r = clamp8(Y + ((91881 * cr + 32768) >> 16));
            const int g = clamp8(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
            const int b = clamp8(Y + ((116130 * cb + 32768) >> 16));
            v = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
        }
    }
    rgb[f.rgb_off + idx] = v | 0xff000000u;
}

#define R_BITS 22
__device__ __forceinline__ int clip22(int s) { return clamp8(s >> R_BITS); }

// horizontal pass: rows [row0, row0 + nrows) x crop columns [crop_x, crop_x + out_w) of the resized width
__global__ __launch_bounds__(256) void resize_h_kernel(const FrameDesc* desc, const uint32_t* rgb, const int* tab, uint32_t* tmp, int out_w) {
    const FrameDesc& f = desc[blockIdx.y];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (!f.need_h || idx >= (long)f.nrows * out_w) return;
    const int ox = (int)(idx % out_w), r = (int)(idx / out_w);
    const uint32_t* line = rgb + f.rgb_off + (long)(f.row0 + r) * f.pw;
    const int xmin = tab[f.bx_off + 2 * ox], cnt = tab[f.bx_off + 2 * ox + 1];
    const int* k = tab + f.kx_off + (long)ox * f.kx;
    int s0 = 1 << (R_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < cnt; ++i) {
        const uint32_t p = line[xmin + i];
        s0 += (int)(p & 255) * k[i]; s1 += (int)((p >> 8) & 255) * k[i]; s2 += (int)((p >> 16) & 255) * k[i];
    }
    tmp[f.tmp_off + idx] = (uint32_t)clip22(s0) | ((uint32_t)clip22(s1) << 8) | ((uint32_t)clip22(s2) << 16);
}

// vertical pass + crop + ToTensor + Normalize: out[c][oy][ox] = (u8 / 255 - mean) / std, three fp32 roundings as torch does
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const FrameDesc* desc, const uint32_t* rgb, const uint32_t* tmp, const int* tab,
                                                            float* out, int out_h, int out_w, float m0, float m1, float m2, float d0, float d1,
                                                            float d2) {
    const FrameDesc& f = desc[blockIdx.y];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)out_h * out_w) return;
    const int ox = (int)(idx % out_w), oy = (int)(idx / out_w);
    // source of the vertical pass: the horizontal-pass buffer (rows relative to row0, columns = crop columns), or the frame itself
    const uint32_t* src = f.need_h ? tmp + f.tmp_off + ox : rgb + f.rgb_off + (long)f.row0 * f.pw + f.crop_x + ox;
    const long pitch = f.need_h ? out_w : f.pw;
    uint32_t p;
    if (f.need_v) {
        const int ymin = tab[f.by_off + 2 * oy] - f.row0, cnt = tab[f.by_off + 2 * oy + 1];
        const int* k = tab + f.ky_off + (long)oy * f.ky;
        int s0 = 1 << (R_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < cnt; ++i) {
            const uint32_t q = src[(long)(ymin + i) * pitch];
            s0 += (int)(q & 255) * k[i]; s1 += (int)((q >> 8) & 255) * k[i]; s2 += (int)((q >> 16) & 255) * k[i];
        }
        p = (uint32_t)clip22(s0) | ((uint32_t)clip22(s1) << 8) | ((uint32_t)clip22(s2) << 16);
    } else {
        p = src[(long)(f.crop_y + oy - f.row0) * pitch];
    }
    float* o = out + f.out_off + idx;
    const long cs = (long)out_h * out_w;
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)(p & 255), 255.f), m0), d0);
    o[cs] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)((p >> 8) & 255), 255.f), m1), d1);
    o[2 * cs] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)((p >> 16) & 255), 255.f), m2), d2);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pillow-style resample coefficients (triangle filter, antialiased): bounds (xmin, count) and 22-bit weights per output pixel
// of [first, first + count_out) out of out_size
// ---------------------------------------------------------------------------------------------------------------------
int resample_ksize(int in_size, int out_size) {
    double scale = (double)in_size / out_size;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil(1.0 * scale) * 2 + 1;
}
void resample_tables(int in_size, int out_size, int first, int count_out, int ksize, int* bounds, int* kk) {
    double filterscale, scale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    std::vector<double> w(ksize);
    for (int o = 0; o < count_out; ++o) {
        const int xx = first + o;
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double v = a < 1.0 ? 1.0 - a : 0.0;
            w[x] = v; ww += v;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) w[x] /= ww;
        for (; x < ksize; ++x) w[x] = 0.0;
        bounds[2 * o] = xmin; bounds[2 * o + 1] = xmax;
        for (x = 0; x < ksize; ++x)
            kk[(long)o * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << R_BITS)) : (int)(0.5 + w[x] * (1 << R_BITS));
    }
}

struct Staging {                       // one pinned host set
    char* host = nullptr; size_t cap = 0;
    hipEvent_t done = nullptr; bool pending = false;
};
struct Decoder {
    int n_threads = 4;
    Staging st[2]; int cur = 0;
    char* dev = nullptr; size_t dev_cap = 0;          // [descs | tables | coefficients] mirror of the staging set
    uint8_t* planes = nullptr; size_t planes_cap = 0;
    uint32_t* rgb = nullptr; size_t rgb_cap = 0;      // pixels
    uint32_t* tmp = nullptr; size_t tmp_cap = 0;      // pixels
    std::vector<FrameDesc> last;                      // host copy of the last batch's descriptors (debug taps)
    std::vector<int> failed;                          // frames of the last lav_decoder_decode call that could not be decoded
};

template <typename T>
bool grow(T*& p, size_t& cap, size_t need) {
    if (need <= cap) return true;
    if (p) { hipDeviceSynchronize(); hipFree(p); p = nullptr; cap = 0; }
    need += need / 4;
    if (hipMalloc((void**)&p, need * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
    cap = need;
    return true;
}
inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }
}  // namespace

extern "C" void* lav_decoder_create(int n_threads) {
    Decoder* d = new Decoder;
    d->n_threads = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&d->st[i].done, hipEventDisableTiming) != hipSuccess) { lav_set_error("lav_decoder_create: hipEventCreate failed"); delete d; return nullptr; }
    return d;
}
extern "C" void lav_decoder_destroy(void* h) {
    Decoder* d = (Decoder*)h;
    if (!d) return;
    hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) { if (d->st[i].host) hipHostFree(d->st[i].host); if (d->st[i].done) hipEventDestroy(d->st[i].done); }
    if (d->dev) hipFree(d->dev);
    if (d->planes) hipFree(d->planes);
    if (d->rgb) hipFree(d->rgb);
    if (d->tmp) hipFree(d->tmp);
    delete d;
}

extern "C" int lav_decoder_decode(void* h, void* stream, int n_frames, const char* const* b64, const long* b64_len, const lav_frame_xform* xf,
                                  int out_h, int out_w, const float* mean3, const float* std3, float* out) {
    Decoder* D = (Decoder*)h;
    if (D) D->failed.clear();                               // before any early return: lav_decoder_failed_frames() describes THIS call only
    LAV_REQUIRE(D && n_frames > 0 && b64 && b64_len && xf && out_h > 0 && out_w > 0 && mean3 && std3 && out, "lav_decoder_decode: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // ---- host stage 1 (threads): base64 + header ------------------------------------------------------------------
    std::vector<std::vector<uint8_t>> jpg(n_frames);
    std::vector<JpegHeader> hdr(n_frames);
    std::vector<std::string> err(n_frames);
    auto parallel = [&](auto&& fn) {
        std::atomic<int> next{0};
        const int nt = D->n_threads < n_frames ? D->n_threads : n_frames;
        std::vector<std::thread> th;
        auto body = [&]() { for (int i; (i = next.fetch_add(1)) < n_frames;) fn(i); };
        for (int t = 1; t < nt; ++t) th.emplace_back(body);
        body();
        for (auto& t : th) t.join();
    };
    parallel([&](int i) {
        if (!b64[i] || b64_len[i] <= 0) { err[i] = "empty frame"; return; }
        jpg[i].resize((size_t)(b64_len[i] / 4 * 3 + 4));
        const long n = b64_decode(b64[i], b64_len[i], jpg[i].data(), (long)jpg[i].size());
        if (n < 0) { err[i] = "base64 decode failed"; return; }
        jpg[i].resize((size_t)n);
        const char* e = jpeg_parse(jpg[i].data(), n, hdr[i], true);
        if (e) err[i] = e;
    });
    auto report = [&]() {                                   // every unreadable frame of the batch, first message
        int first = -1;
        for (int i = 0; i < n_frames; ++i)
            if (!err[i].empty()) { D->failed.push_back(i); if (first < 0) first = i; }
        if (first >= 0) lav_set_error("lav_decoder_decode: frame %d: %s (%d unreadable frame(s) in the batch)", first, err[first].c_str(), (int)D->failed.size());
        return first >= 0;
    };
    if (report()) return LAV_E_ARG;
    // ---- layout of this batch --------------------------------------------------------------------------------------
    std::vector<FrameDesc> desc(n_frames);
    size_t n_tab = 0, n_coef = 0, n_plane = 0, n_rgb = 0, n_tmp = 0;
    long max_blocks = 0, max_pix = 0, max_hrows = 0;
    for (int i = 0; i < n_frames; ++i) {
        FrameDesc& f = desc[i]; const JpegHeader& H = hdr[i]; const lav_frame_xform& x = xf[i];
        memset(&f, 0, sizeof f);
        f.w = H.w; f.h = H.h; f.ncomp = H.ncomp;
        f.sub = H.ncomp == 1 ? 0 : (H.hs[0] == 2 ? (H.vs[0] == 2 ? 2 : 1) : 0);
        f.cw = f.sub ? (H.w + 1) / 2 : H.w; f.ch = f.sub == 2 ? (H.h + 1) / 2 : H.h;
        long blocks = 0;
        for (int c = 0; c < H.ncomp; ++c) {
            f.bw[c] = H.bw[c]; f.bh[c] = H.bh[c];
            f.coef_off[c] = (long)n_coef; n_coef += (size_t)H.bw[c] * H.bh[c] * 64;
            f.plane_off[c] = (long)n_plane; n_plane += up16((size_t)H.bw[c] * H.bh[c] * 64);
            memcpy(f.q[c], H.q[H.tq[c]], sizeof f.q[c]);
            blocks += (long)H.bw[c] * H.bh[c];
        }
        LAV_REQUIRE(x.pad_left >= 0 && x.pad_top >= 0, "lav_decoder_decode: frame %d: negative padding", i);
        f.pad_l = x.pad_left; f.pad_t = x.pad_top; f.pw = H.w + 2 * x.pad_left; f.ph = H.h + 2 * x.pad_top;
        f.rw = x.resize_w; f.rh = x.resize_h;
        LAV_REQUIRE(f.rw > 0 && f.rh > 0 && x.crop_x >= 0 && x.crop_y >= 0 && x.crop_x + out_w <= f.rw && x.crop_y + out_h <= f.rh,
                    "lav_decoder_decode: frame %d: crop window (%d,%d)+(%d,%d) outside the resized frame %dx%d", i, x.crop_x, x.crop_y, out_w, out_h, f.rw, f.rh);
        LAV_REQUIRE(x.out_index >= 0, "lav_decoder_decode: frame %d: negative output slot", i);
        f.crop_x = x.crop_x; f.crop_y = x.crop_y;
        f.need_h = f.rw != f.pw; f.need_v = f.rh != f.ph;
        f.kx = f.need_h ? resample_ksize(f.pw, f.rw) : 0; f.ky = f.need_v ? resample_ksize(f.ph, f.rh) : 0;
        f.bx_off = (long)n_tab; n_tab += f.need_h ? 2 * (size_t)out_w : 0;
        f.kx_off = (long)n_tab; n_tab += (size_t)f.kx * out_w;
        f.by_off = (long)n_tab; n_tab += f.need_v ? 2 * (size_t)out_h : 0;
        f.ky_off = (long)n_tab; n_tab += (size_t)f.ky * out_h;
        f.rgb_off = (long)n_rgb; n_rgb += (size_t)f.pw * f.ph;
        f.out_off = x.out_index * 3L * out_h * out_w;
        if (blocks > max_blocks) max_blocks = blocks;
        if ((long)f.pw * f.ph > max_pix) max_pix = (long)f.pw * f.ph;
    }
    const size_t off_tab = up16(sizeof(FrameDesc) * n_frames);
    const size_t off_coef = up16(off_tab + n_tab * sizeof(int));
    const size_t total = off_coef + n_coef * sizeof(int16_t);
    // ---- staging set --------------------------------------------------------------------------------------------------
    Staging& S = D->st[D->cur]; D->cur ^= 1;
    if (S.pending) { hipEventSynchronize(S.done); S.pending = false; }
    if (S.cap < total) {
        if (S.host) hipHostFree(S.host);
        S.cap = total + total / 4;
        if (hipHostMalloc((void**)&S.host, S.cap, hipHostMallocDefault) != hipSuccess) { S.host = nullptr; S.cap = 0; lav_set_error("lav_decoder_decode: pinned allocation of %zu bytes failed", total); return LAV_E_LAUNCH; }
    }
    int* tab = (int*)(S.host + off_tab);
    int16_t* coef = (int16_t*)(S.host + off_coef);
    // ---- host stage 2 (threads): entropy decode + resample tables -----------------------------------------------------
    parallel([&](int i) {
        FrameDesc& f = desc[i];
        const char* e = jpeg_entropy_decode(jpg[i].data(), (long)jpg[i].size(), hdr[i], coef + f.coef_off[0]);
        if (e) { err[i] = e; return; }
        if (f.need_v) resample_tables(f.ph, f.rh, f.crop_y, out_h, f.ky, tab + f.by_off, tab + f.ky_off);
        if (f.need_h) resample_tables(f.pw, f.rw, f.crop_x, out_w, f.kx, tab + f.bx_off, tab + f.kx_off);
    });
    if (report()) return LAV_E_ARG;
    for (int i = 0; i < n_frames; ++i) {                    // rows the vertical pass reads
        FrameDesc& f = desc[i];
        if (f.need_v) {
            const int* b = tab + f.by_off;
            f.row0 = b[0]; f.nrows = b[2 * (out_h - 1)] + b[2 * (out_h - 1) + 1] - b[0];
        } else { f.row0 = f.crop_y; f.nrows = out_h; }
        f.tmp_off = (long)n_tmp; n_tmp += f.need_h ? (size_t)f.nrows * out_w : 0;
        if (f.need_h && f.nrows > max_hrows) max_hrows = f.nrows;
    }
    memcpy(S.host, desc.data(), sizeof(FrameDesc) * n_frames);
    // ---- device stage -------------------------------------------------------------------------------------------------
    if (!grow(D->dev, D->dev_cap, total) || !grow(D->planes, D->planes_cap, n_plane) || !grow(D->rgb, D->rgb_cap, n_rgb) ||
        !grow(D->tmp, D->tmp_cap, n_tmp ? n_tmp : 1)) { lav_set_error("lav_decoder_decode: device allocation failed"); return LAV_E_LAUNCH; }
    if (hipMemcpyAsync(D->dev, S.host, total, hipMemcpyHostToDevice, s) != hipSuccess) { lav_set_error("lav_decoder_decode: host-to-device copy failed"); return LAV_E_LAUNCH; }
    hipEventRecord(S.done, s); S.pending = true;
    const FrameDesc* dd = (const FrameDesc*)D->dev;
    const int* dtab = (const int*)(D->dev + off_tab);
    const int16_t* dcoef = (const int16_t*)(D->dev + off_coef);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 63) / 64), n_frames), dim3(64), 0, s, dd, dcoef, D->planes);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((max_pix + 255) / 256), n_frames), dim3(256), 0, s, dd, (const uint8_t*)D->planes, D->rgb);
    if (max_hrows)
        hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((max_hrows * out_w + 255) / 256), n_frames), dim3(256), 0, s, dd, (const uint32_t*)D->rgb, dtab, D->tmp, out_w);
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((unsigned)(((long)out_h * out_w + 255) / 256), n_frames), dim3(256), 0, s, dd, (const uint32_t*)D->rgb,
                       (const uint32_t*)D->tmp, dtab, out, out_h, out_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    D->last = desc;
    return lav_check_launch("lav_decoder_decode");
}

extern "C" int lav_decoder_failed_frames(void* h, int* frames, int capacity) {
    Decoder* D = (Decoder*)h;
    if (!D) return LAV_E_ARG;
    const int n = (int)D->failed.size();
    for (int i = 0; i < n && i < capacity; ++i) frames[i] = D->failed[i];
    return n;
}

extern "C" int lav_decoder_read_rgb(void* h, int frame, uint8_t* rgb, long capacity, int* w, int* hh) {
    Decoder* D = (Decoder*)h;
    LAV_REQUIRE(D && frame >= 0 && frame < (int)D->last.size() && rgb && w && hh, "lav_decoder_read_rgb: bad arguments");
    const FrameDesc& f = D->last[frame];
    const long n = (long)f.pw * f.ph;
    LAV_REQUIRE(capacity >= 3 * n, "lav_decoder_read_rgb: buffer too small (%ld < %ld)", capacity, 3 * n);
    std::vector<uint32_t> px((size_t)n);
    hipDeviceSynchronize();
    if (hipMemcpy(px.data(), D->rgb + f.rgb_off, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { lav_set_error("lav_decoder_read_rgb: copy failed"); return LAV_E_LAUNCH; }
    for (long i = 0; i < n; ++i) { rgb[3 * i] = px[i] & 255; rgb[3 * i + 1] = (px[i] >> 8) & 255; rgb[3 * i + 2] = (px[i] >> 16) & 255; }
    *w = f.pw; *hh = f.ph;
    return 0;
}
