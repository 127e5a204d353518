// pngio.cpp — the on-disk frame format of the path: cb_render/cb_rgb_%04d.png and best_render.png
// (reference reconstruction/combined_rendering.py:157-159 writes them with cv2.imwrite one by one inside the
// render loop; clip_scoring.py:89-104 reads them back with use_cache_renders, :222-223 writes best_render.png).
// Host code only: 8-bit RGB PNG encode / decode on zlib, and a pool of worker threads that turns the frames a
// render-and-score pass streams back into files while the GPU works on the next chunk.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "d2r_internal.h"
#include "pngio.h"

namespace {

void put32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t> &out, const char type[4], const uint8_t *data, size_t n)
{
    put32(out, (uint32_t)n);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

// ---- a zlib stream of ONE dynamic-Huffman deflate block with literals only (no matches) ----
// What zlib's Z_HUFFMAN_ONLY strategy produces, written for speed: a histogram pass, length-limited Huffman codes, and an emit
// loop on a 64-bit bit buffer (zlib spends its time in per-byte tallying and block bookkeeping: 211 MB/s per thread on Sub-filtered
// frames, tools/png_strategy_probe.py).  Any inflater reads it.  `out` must hold n + n / 8 + 1024 bytes.
struct HuffCode { uint16_t code; uint8_t len; };

// code lengths (<= 15) for 257 symbols from their counts: Huffman's algorithm on a two-queue merge of the sorted counts; when the
// deepest leaf is too deep the counts are halved (rounded up) and the tree is rebuilt — rare (a 16-level tree needs counts that grow
// like Fibonacci numbers) and costs microseconds
void huff_lengths(const uint32_t *cnt_in, int n_sym, uint8_t *len)
{
    std::vector<uint32_t> cnt(cnt_in, cnt_in + n_sym);
    for (;;) {
        struct Node { uint64_t w; int l, r; };
        std::vector<Node> nodes;
        std::vector<int> leaves;
        for (int i = 0; i < n_sym; i++)
            if (cnt[i]) leaves.push_back(i);
        for (int i = 0; i < n_sym; i++) len[i] = 0;
        if (leaves.size() == 1) { len[leaves[0]] = 1; return; }
        std::sort(leaves.begin(), leaves.end(), [&](int a, int b) { return cnt[a] != cnt[b] ? cnt[a] < cnt[b] : a < b; });
        nodes.reserve(2 * leaves.size());
        for (int s : leaves) nodes.push_back({cnt[s], -1 - s, 0});
        size_t qa = 0, qb = leaves.size(), n_leaf = leaves.size();      // queue a: leaves, queue b: internal nodes (created in order of weight)
        auto pop = [&]() {
            const bool from_a = qa < n_leaf && (qb >= nodes.size() || nodes[qa].w <= nodes[qb].w);
            return (int)(from_a ? qa++ : qb++);
        };
        while ((n_leaf - qa) + (nodes.size() - qb) > 1) {
            const int x = pop(), y = pop();
            nodes.push_back({nodes[x].w + nodes[y].w, x, y});
        }
        // depths from the root down (children precede parents in `nodes`)
        std::vector<uint8_t> depth(nodes.size(), 0);
        int deepest = 0;
        for (size_t i = nodes.size(); i-- > n_leaf;) {
            depth[nodes[i].l] = depth[nodes[i].r] = (uint8_t)(depth[i] + 1);
        }
        for (size_t i = 0; i < n_leaf; i++) {
            len[-1 - nodes[i].l] = depth[i];
            deepest = std::max<int>(deepest, depth[i]);
        }
        if (deepest <= 15) return;
        for (int i = 0; i < n_sym; i++)
            if (cnt[i]) cnt[i] = (cnt[i] + 1) >> 1;
    }
}

size_t deflate_huffman_only(const uint8_t *in, size_t n, uint8_t *out)
{
    uint32_t h4[4][256];
    memset(h4, 0, sizeof h4);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {            // four tables: consecutive equal bytes do not wait for one another's increment
        h4[0][in[i]]++; h4[1][in[i + 1]]++; h4[2][in[i + 2]]++; h4[3][in[i + 3]]++;
    }
    for (; i < n; i++) h4[0][in[i]]++;
    uint32_t cnt[257];
    for (int s = 0; s < 256; s++) cnt[s] = h4[0][s] + h4[1][s] + h4[2][s] + h4[3][s];
    cnt[256] = 1;                            // end of block
    uint8_t len[257];
    huff_lengths(cnt, 257, len);
    // canonical codes, bit-reversed (deflate sends Huffman codes most significant bit first into an LSB-first stream)
    uint16_t next[16] = {0}, bl_count[16] = {0};
    for (int s = 0; s < 257; s++) bl_count[len[s]]++;
    bl_count[0] = 0;
    uint16_t code = 0;
    for (int b = 1; b <= 15; b++) {
        code = (uint16_t)((code + bl_count[b - 1]) << 1);
        next[b] = code;
    }
    HuffCode hc[257];
    for (int s = 0; s < 257; s++) {
        uint16_t c = len[s] ? next[len[s]]++ : 0, r = 0;
        for (int b = 0; b < len[s]; b++) r = (uint16_t)((r << 1) | ((c >> b) & 1));
        hc[s] = {r, len[s]};
    }
    uint8_t *o = out;
    *o++ = 0x78;
    *o++ = 0x01;
    uint64_t bits = 0;
    int nb = 0;
    auto put = [&](uint32_t v, int k) {
        bits |= (uint64_t)v << nb;
        nb += k;
        if (nb >= 32) {
            memcpy(o, &bits, 4);             // little-endian hosts (x86-64)
            o += 4;
            bits >>= 32;
            nb -= 32;
        }
    };
    // block header: final, dynamic; 257 literal/length codes, 1 distance code; the code-length alphabet is sent flat — symbols 0..15 in
    // four bits each (a complete code), 16..18 unused — so a code length costs four bits: 129 bytes per frame
    put(1, 1);
    put(2, 2);
    put(0, 5);                               // HLIT: 257
    put(0, 5);                               // HDIST: 1
    put(15, 4);                              // HCLEN: 19
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int k = 0; k < 19; k++) put(order[k] < 16 ? 4 : 0, 3);
    auto rev4 = [](uint32_t v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); };     // canonical 4-bit code of symbol v = v
    for (int s = 0; s < 257; s++) put(rev4(len[s]), 4);
    put(rev4(0), 4);                         // the one distance code: length 0 (no distances are used)
    for (i = 0; i < n; i++) put(hc[in[i]].code, hc[in[i]].len);
    put(hc[256].code, hc[256].len);
    while (nb > 0) {
        *o++ = (uint8_t)bits;
        bits >>= 8;
        nb -= 8;
    }
    const uint32_t ad = (uint32_t)adler32(adler32(0L, Z_NULL, 0), in, (uInt)n);
    *o++ = (uint8_t)(ad >> 24); *o++ = (uint8_t)(ad >> 16); *o++ = (uint8_t)(ad >> 8); *o++ = (uint8_t)ad;
    return (size_t)(o - out);
}

}  // namespace

// ---- frames that differ from a known BACKGROUND frame in a few rows (what a render-and-score pass streams out: the candidate's object
// covers a band of the frame, every other scanline is the background's) ----
// The background's Sub-filtered scanlines are Huffman-coded ONCE, with one code shared by all frames (the background's histogram, every
// literal kept alive); a frame is then written as: the shared block header, and per scanline either the background's ready-made bit
// string appended at the current bit position (a shift-and-copy at ~GB/s) or, where the scanline differs, a fresh filter + emit pass.
// Adler-32 of the filtered bytes is combined from per-scanline values (adler32_combine), only the CRC-32 of the compressed bytes is
// computed per frame.  Any inflater reads the result; pixels are identical to d2r_png_encode's (PNG is lossless), files are a few
// per cent larger where the object's statistics differ from the background's.
struct D2rPngBase {
    uint32_t w = 0, h = 0;
    size_t row = 0;                          // 3 w
    std::vector<uint8_t> rgb;                // the background frame
    HuffCode hc[257];
    std::vector<uint8_t> head;               // zlib header + deflate block header, as a bit string (starts at bit 0)
    size_t head_bits = 0;
    std::vector<uint8_t> bits;               // per-scanline bit strings, each starting at a byte boundary
    std::vector<size_t> row_off;             // byte offset of scanline y in `bits`
    std::vector<uint32_t> row_nbits;
    std::vector<uint32_t> row_adler;         // adler32 of the scanline's filtered bytes (filter type byte included), standalone
};

namespace {

struct BitWriter {
    uint8_t *o;
    uint64_t acc = 0;
    int nb = 0;                              // < 32 between calls
    explicit BitWriter(uint8_t *out) : o(out) {}
    inline void put(uint32_t v, int k)       // k <= 32, v < 2^k
    {
        acc |= (uint64_t)v << nb;
        nb += k;
        if (nb >= 32) {
            memcpy(o, &acc, 4);              // little-endian hosts (x86-64)
            o += 4;
            acc >>= 32;
            nb -= 32;
        }
    }
    void append(const uint8_t *src, size_t nbits)      // a bit string that starts at bit 0 of src[0]
    {
        size_t i = 0;
        for (; i + 32 <= nbits; i += 32) {
            uint32_t v;
            memcpy(&v, src + (i >> 3), 4);
            put(v, 32);
        }
        size_t left = nbits - i;
        const uint8_t *p = src + (i >> 3);
        while (left >= 8) {
            put(*p++, 8);
            left -= 8;
        }
        if (left) put(*p & ((1u << left) - 1u), (int)left);
    }
    size_t finish(uint8_t *base)             // pad to a byte, return the byte count
    {
        while (nb > 0) {
            *o++ = (uint8_t)acc;
            acc >>= 8;
            nb -= 8;
        }
        nb = 0;
        return (size_t)(o - base);
    }
};

inline void sub_filter_row(const uint8_t *src, size_t row, uint8_t *dst)      // dst: row + 1 bytes
{
    dst[0] = 1;
    dst[1] = src[0]; dst[2] = src[1]; dst[3] = src[2];
    for (size_t i = 3; i < row; i++) dst[1 + i] = (uint8_t)(src[i] - src[i - 3]);
}

void canonical_codes(const uint8_t *len, HuffCode *hc)
{
    uint16_t next[16] = {0}, bl_count[16] = {0};
    for (int s = 0; s < 257; s++) bl_count[len[s]]++;
    bl_count[0] = 0;
    uint16_t code = 0;
    for (int b = 1; b <= 15; b++) {
        code = (uint16_t)((code + bl_count[b - 1]) << 1);
        next[b] = code;
    }
    for (int s = 0; s < 257; s++) {
        uint16_t c = len[s] ? next[len[s]]++ : 0, r = 0;
        for (int b = 0; b < len[s]; b++) r = (uint16_t)((r << 1) | ((c >> b) & 1));
        hc[s] = {r, len[s]};
    }
}

}  // namespace

std::shared_ptr<const D2rPngBase> d2r_png_base_build(const uint8_t *bg_rgb, uint32_t w, uint32_t h)
{
    if (!bg_rgb || w == 0 || h == 0 || w > 32768 || h > 32768) return nullptr;
    auto B = std::make_shared<D2rPngBase>();
    B->w = w;
    B->h = h;
    B->row = (size_t)w * 3;
    const size_t row = B->row;
    B->rgb.assign(bg_rgb, bg_rgb + row * h);
    std::vector<uint8_t> f((row + 1) * h);
    uint32_t cnt[257];
    for (int s2 = 0; s2 < 257; s2++) cnt[s2] = 1;                 // every literal (and the end-of-block symbol) keeps a code: the object's bytes are not known yet
    for (uint32_t y = 0; y < h; y++) {
        sub_filter_row(bg_rgb + row * y, row, &f[(row + 1) * y]);
        for (size_t i = 0; i <= row; i++) cnt[f[(row + 1) * y + i]] += 4;   // (x 4: the floor of 1 stays small beside real counts)
    }
    uint8_t len[257];
    huff_lengths(cnt, 257, len);
    canonical_codes(len, B->hc);
    // zlib header + block header, exactly as deflate_huffman_only writes them
    B->head.assign(256, 0);
    {
        BitWriter bw(B->head.data());
        bw.put(0x78, 8);
        bw.put(0x01, 8);
        bw.put(1, 1);
        bw.put(2, 2);
        bw.put(0, 5);
        bw.put(0, 5);
        bw.put(15, 4);
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int k = 0; k < 19; k++) bw.put(order[k] < 16 ? 4 : 0, 3);
        auto rev4 = [](uint32_t v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); };
        for (int s2 = 0; s2 < 257; s2++) bw.put(rev4(len[s2]), 4);
        bw.put(rev4(0), 4);
        B->head_bits = 16 + 3 + 5 + 5 + 4 + 19 * 3 + 258 * 4;
        bw.finish(B->head.data());
    }
    // per-scanline bit strings (worst case 15 bits per byte)
    B->row_off.resize(h);
    B->row_nbits.resize(h);
    B->row_adler.resize(h);
    std::vector<uint8_t> tmp((row + 1) * 2 + 16);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *fr = &f[(row + 1) * y];
        std::fill(tmp.begin(), tmp.end(), 0);
        BitWriter bw(tmp.data());
        size_t nbits = 0;
        for (size_t i = 0; i <= row; i++) {
            bw.put(B->hc[fr[i]].code, B->hc[fr[i]].len);
            nbits += B->hc[fr[i]].len;
        }
        const size_t nbytes = bw.finish(tmp.data());
        B->row_off[y] = B->bits.size();
        B->row_nbits[y] = (uint32_t)nbits;
        B->bits.insert(B->bits.end(), tmp.begin(), tmp.begin() + nbytes);
        B->bits.insert(B->bits.end(), 4, 0);                     // slack: append() reads whole 32-bit words
        B->row_adler[y] = (uint32_t)adler32(adler32(0L, Z_NULL, 0), fr, (uInt)(row + 1));
    }
    return B;
}

int d2r_png_encode_delta(const D2rPngBase &B, const uint8_t *rgb, std::vector<uint8_t> &out, std::string &err)
{
    if (!rgb) {
        err = "null frame";
        return D2R_ERR_INVALID;
    }
    const size_t row = B.row;
    thread_local std::vector<uint8_t> z, frow;
    const size_t cap = (row + 1) * B.h * 2 + 2048;               // 15-bit codes at worst
    if (z.size() < cap) z.resize(cap);
    if (frow.size() < row + 1) frow.resize(row + 1);
    BitWriter bw(z.data());
    bw.append(B.head.data(), B.head_bits);
    uLong ad = adler32(0L, Z_NULL, 0);
    for (uint32_t y = 0; y < B.h; y++) {
        const uint8_t *src = rgb + row * y;
        if (memcmp(src, &B.rgb[row * y], row) == 0) {
            bw.append(&B.bits[B.row_off[y]], B.row_nbits[y]);
            ad = adler32_combine(ad, B.row_adler[y], (z_off_t)(row + 1));
        } else {
            sub_filter_row(src, row, frow.data());
            for (size_t i = 0; i <= row; i++) bw.put(B.hc[frow[i]].code, B.hc[frow[i]].len);
            ad = adler32(ad, frow.data(), (uInt)(row + 1));
        }
    }
    bw.put(B.hc[256].code, B.hc[256].len);
    size_t n = bw.finish(z.data());
    z[n++] = (uint8_t)(ad >> 24); z[n++] = (uint8_t)(ad >> 16); z[n++] = (uint8_t)(ad >> 8); z[n++] = (uint8_t)ad;
    out.clear();
    out.reserve(n + 64);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, B.w);
    put32(ihdr, B.h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), n);
    chunk(out, "IEND", nullptr, 0);
    return D2R_OK;
}

int d2r_png_write_file_delta(const D2rPngBase &B, const uint8_t *rgb, const std::string &path, std::string &err)
{
    thread_local std::vector<uint8_t> bytes;
    int rc = d2r_png_encode_delta(B, rgb, bytes, err);
    if (rc) return rc;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        err = "cannot open " + path + " for writing";
        return D2R_ERR_INVALID;
    }
    const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (fclose(f) != 0 || !ok) {
        err = "short write to " + path;
        return D2R_ERR_INVALID;
    }
    return D2R_OK;
}

// One RGB frame -> PNG bytes: colour type 2, bit depth 8, no interlace (any decoder returns the same pixels: PNG is lossless).
int d2r_png_encode(const uint8_t *rgb, uint32_t w, uint32_t h, int level, std::vector<uint8_t> &out, std::string &err)
{
    if (!rgb || w == 0 || h == 0 || w > 32768 || h > 32768) {
        err = "bad image size";
        return D2R_ERR_INVALID;
    }
    const size_t row = (size_t)w * 3;
    // scratch buffers live as long as the worker thread: three fresh allocations of a third of a megabyte per frame are
    // mmap / munmap + page faults in glibc, and with many threads those serialise on the process' address-space lock —
    // measured on the 256-core MI355X host: 3.3 k files/s with 16 threads, 2.3 k with 64, 1.7 k with 128
    thread_local std::vector<uint8_t> raw, z;
    const size_t raw_n = (row + 1) * h;
    if (raw.size() < raw_n) raw.resize(raw_n);
    // level < 0 (the default): filter type 1 (Sub: each byte minus the same channel of the pixel to its left, cv2.imwrite's filter —
    // the reference's writer, combined_rendering.py:157-159) + Huffman-only deflate (deflate_huffman_only above).  On rendered frames (tools/png_strategy_probe.py,
    // 640x360, one thread of the MI355X host) zlib's Huffman-only strategy gives 190 KiB per frame at 211 MB/s, cv2's run-length strategy
    // the same size at 142 MB/s (textured NeRF backgrounds have no runs to find), unfiltered scanlines at level 1 305 KiB at 109 MB/s.
    // level 0..9: unfiltered scanlines, default strategy, that level.
    const bool fast = level < 0;
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *dst = &raw[(row + 1) * y];
        const uint8_t *src = rgb + row * y;
        if (fast) {
            dst[0] = 1;
            dst[1] = src[0]; dst[2] = src[1]; dst[3] = src[2];
            for (size_t i = 3; i < row; i++) dst[1 + i] = (uint8_t)(src[i] - src[i - 3]);
        } else {
            dst[0] = 0;
            memcpy(dst + 1, src, row);
        }
    }
    uLongf cap = fast ? (uLongf)(raw_n + raw_n / 8 + 1024) : compressBound((uLong)raw_n);
    if (z.size() < cap) z.resize(cap);
    if (fast) cap = (uLongf)deflate_huffman_only(raw.data(), raw_n, z.data());
    else {
    // one deflate state per worker thread too (compress2 would allocate and free its quarter megabyte per frame)
    struct Deflater {
        z_stream zs;
        bool live = false;
        int level = -2;
        ~Deflater() { if (live) deflateEnd(&zs); }
    };
    thread_local Deflater df;
    const int lv = std::min(level, 9);
    if (!df.live || df.level != lv) {
        if (df.live) deflateEnd(&df.zs);
        memset(&df.zs, 0, sizeof df.zs);
        df.live = deflateInit(&df.zs, lv) == Z_OK;
        df.level = lv;
    } else if (deflateReset(&df.zs) != Z_OK) {
        deflateEnd(&df.zs);
        df.live = false;
    }
    if (!df.live) {
        err = "zlib deflateInit failed";
        return D2R_ERR_MEMORY;
    }
    df.zs.next_in = raw.data();
    df.zs.avail_in = (uInt)raw_n;
    df.zs.next_out = z.data();
    df.zs.avail_out = (uInt)cap;
    if (deflate(&df.zs, Z_FINISH) != Z_STREAM_END) {
        err = "zlib deflate failed";
        return D2R_ERR_MEMORY;
    }
    cap = (uLongf)(cap - df.zs.avail_out);
    }
    out.clear();
    out.reserve(cap + 64);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, w);
    put32(ihdr, h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), cap);
    chunk(out, "IEND", nullptr, 0);
    return D2R_OK;
}

int d2r_png_write_file(const uint8_t *rgb, uint32_t w, uint32_t h, int level, const std::string &path, std::string &err)
{
    thread_local std::vector<uint8_t> bytes;
    int rc = d2r_png_encode(rgb, w, h, level, bytes, err);
    if (rc) return rc;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        err = "cannot open " + path + " for writing";
        return D2R_ERR_INVALID;
    }
    const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (fclose(f) != 0 || !ok) {
        err = "short write to " + path;
        return D2R_ERR_INVALID;
    }
    return D2R_OK;
}

// PNG bytes -> RGB.  Reads what PNG writers produce for 8-bit images without a palette: grey, grey + alpha, RGB,
// RGBA (alpha dropped, grey replicated), all five scanline filters, IDAT split over any number of chunks.  Interlaced,
// 16-bit and palette images are refused with a message (cv2.imwrite / PIL never write them for uint8 RGB arrays).
int d2r_png_decode(const uint8_t *p, size_t n, uint32_t want_w, uint32_t want_h, uint8_t *rgb_out, uint32_t *w_out,
                   uint32_t *h_out, std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!p || n < 8 + 25 || memcmp(p, sig, 8)) {
        err = "not a PNG file";
        return D2R_ERR_INVALID;
    }
    auto rd32 = [&](size_t at) { return ((uint32_t)p[at] << 24) | ((uint32_t)p[at + 1] << 16) | ((uint32_t)p[at + 2] << 8) | p[at + 3]; };
    size_t at = 8;
    uint32_t w = 0, h = 0, ch = 0;
    std::vector<uint8_t> z;
    bool have_hdr = false, done = false;
    while (!done && at + 12 <= n) {
        const uint32_t len = rd32(at);
        if ((size_t)len > n - at - 12) {
            err = "truncated PNG chunk";
            return D2R_ERR_INVALID;
        }
        const uint8_t *type = p + at + 4, *data = p + at + 8;
        if (rd32(at + 8 + len) != (uint32_t)crc32(0L, type, (uInt)(len + 4))) {
            err = "PNG chunk CRC mismatch";
            return D2R_ERR_INVALID;
        }
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) {
                err = "bad IHDR";
                return D2R_ERR_INVALID;
            }
            w = rd32(at + 8);
            h = rd32(at + 12);
            const uint8_t depth = data[8], ctype = data[9], interlace = data[12];
            if (depth != 8 || interlace != 0 || (ctype != 0 && ctype != 2 && ctype != 4 && ctype != 6)) {
                err = "unsupported PNG (need 8-bit grey / RGB with or without alpha, not interlaced)";
                return D2R_ERR_UNSUPPORTED;
            }
            ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : 4;
            have_hdr = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            z.insert(z.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        }
        at += 12 + (size_t)len;
    }
    if (!have_hdr || !done || w == 0 || h == 0 || w > 32768 || h > 32768) {
        err = "incomplete PNG";
        return D2R_ERR_INVALID;
    }
    if (w_out) *w_out = w;
    if (h_out) *h_out = h;
    if (!rgb_out) return D2R_OK;
    if ((want_w && want_w != w) || (want_h && want_h != h)) {
        err = "PNG is " + std::to_string(w) + "x" + std::to_string(h) + ", expected " + std::to_string(want_w) + "x" + std::to_string(want_h);
        return D2R_ERR_INVALID;
    }
    const size_t row = (size_t)w * ch;
    std::vector<uint8_t> raw((row + 1) * h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, z.data(), (uLong)z.size()) != Z_OK || got != raw.size()) {
        err = "PNG pixel data does not inflate to the size its header announces";
        return D2R_ERR_INVALID;
    }
    std::vector<uint8_t> prev(row, 0), cur(row);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t ft = raw[(row + 1) * y];
        const uint8_t *src = &raw[(row + 1) * y + 1];
        for (size_t i = 0; i < row; i++) {
            const int a = i >= ch ? cur[i - ch] : 0, b = prev[i], c = i >= ch ? prev[i - ch] : 0;
            int pred;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default:
                err = "bad PNG scanline filter";
                return D2R_ERR_INVALID;
            }
            cur[i] = (uint8_t)(src[i] + pred);
        }
        uint8_t *dst = rgb_out + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *q = &cur[(size_t)x * ch];
            if (ch <= 2) dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = q[0];
            else { dst[3 * x] = q[0]; dst[3 * x + 1] = q[1]; dst[3 * x + 2] = q[2]; }
        }
        prev.swap(cur);
    }
    return D2R_OK;
}

int d2r_png_read_file(const std::string &path, uint32_t want_w, uint32_t want_h, uint8_t *rgb_out, uint32_t *w_out,
                      uint32_t *h_out, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return D2R_ERR_INVALID;
    }
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + k);
    fclose(f);
    int rc = d2r_png_decode(bytes.data(), bytes.size(), want_w, want_h, rgb_out, w_out, h_out, err);
    if (rc) err = path + ": " + err;
    return rc;
}

std::string d2r_png_name(const std::string &dir, uint32_t index)
{
    char name[64];
    snprintf(name, sizeof name, "cb_rgb_%04u.png", index);         // the reference's '%04d' (combined_rendering.py:159)
    return dir.empty() ? std::string(name) : dir + "/" + name;
}

// ------------------------------------------------------------------ worker pool

struct D2rJobPool::Impl {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::pair<int, std::function<int(std::string &)>>> q;     // (group, job)
    size_t pending[D2R_POOL_GROUPS] = {};
    bool stop = false;
    int first_rc = 0;
    std::string first_err;

    void work()
    {
        for (;;) {
            std::pair<int, std::function<int(std::string &)>> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                job = std::move(q.front());
                q.pop_front();
            }
            std::string err;
            int rc;
            try {
                rc = job.second(err);
            } catch (const std::exception &e) {
                rc = D2R_ERR_MEMORY;
                err = e.what();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc && !first_rc) {
                    first_rc = rc;
                    first_err = err;
                }
                pending[job.first]--;
            }
            cv_idle.notify_all();
        }
    }
};

D2rJobPool::D2rJobPool(int n_threads) : impl(new Impl())
{
    n = std::max(1, n_threads);
    for (int i = 0; i < n; i++) impl->threads.emplace_back([this] { impl->work(); });
}

D2rJobPool::~D2rJobPool()
{
    {
        std::lock_guard<std::mutex> lk(impl->mu);
        impl->stop = true;
    }
    impl->cv_work.notify_all();
    for (auto &t : impl->threads) t.join();
    delete impl;
}

void D2rJobPool::submit(int group, std::function<int(std::string &)> job)
{
    {
        std::lock_guard<std::mutex> lk(impl->mu);
        impl->pending[group % D2R_POOL_GROUPS]++;
        impl->q.emplace_back(group % D2R_POOL_GROUPS, std::move(job));
    }
    impl->cv_work.notify_one();
}

void D2rJobPool::wait(int group)
{
    std::unique_lock<std::mutex> lk(impl->mu);
    impl->cv_idle.wait(lk, [&] {
        if (group >= 0) return impl->pending[group % D2R_POOL_GROUPS] == 0;
        for (size_t p : impl->pending)
            if (p) return false;
        return true;
    });
}

int D2rJobPool::take_error(std::string &err)
{
    std::lock_guard<std::mutex> lk(impl->mu);
    const int rc = impl->first_rc;
    err = impl->first_err;
    impl->first_rc = 0;
    impl->first_err.clear();
    return rc;
}

// Worker threads for frame / text IO: the CPUs the process may really use — hardware threads capped by the container's CPU
// quota (cgroup v2 cpu.max, v1 cfs quota).  Measured on a 256-core MI355X host whose container is granted 16 CPUs: PNG
// encoding runs 3.1 k files/s with 16 threads, 1.9 k with 64, 1.5 k with 128 (throttled threads hold the others up).
int d2r_default_io_threads()
{
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 4;
    auto read2 = [](const char *path, long long &a, long long &b) {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char t0[32] = {0}, t1[32] = {0};
        const int got = fscanf(f, "%31s %31s", t0, t1);
        fclose(f);
        if (got < 1 || !strcmp(t0, "max")) return false;
        a = atoll(t0);
        b = got >= 2 ? atoll(t1) : 0;
        return true;
    };
    long long q = 0, p = 0;
    if (read2("/sys/fs/cgroup/cpu.max", q, p) && q > 0 && p > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / p));
    else if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, p) && q > 0) {
        long long per = 0, dummy = 0;
        if (read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", per, dummy) && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / per));
    }
    // one process per GPU: the ranks of a node share those CPUs (torch.distributed.run exports LOCAL_WORLD_SIZE; D2R_LOCAL_WORLD_SIZE
    // overrides it for other launchers) — eight ranks that each started the whole quota's worth of workers would throttle one another
    for (const char *key : {"D2R_LOCAL_WORLD_SIZE", "LOCAL_WORLD_SIZE"})
        if (const char *v = getenv(key)) {
            const long long lws = atoll(v);
            if (lws > 1) n = (unsigned)std::max<long long>(1, (long long)n / lws);
            break;
        }
    return (int)std::min(64u, std::max(1u, n));
}

// ------------------------------------------------------------------ C ABI

extern "C" {

int d2r_png_write(const uint8_t *rgb, uint32_t w, uint32_t h, const char *path, int level)
{
    if (!rgb || !path) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    std::string err;
    int rc = d2r_png_write_file(rgb, w, h, level, path, err);
    return rc ? d2r_fail(nullptr, rc, err) : D2R_OK;
}

int d2r_png_write_batch(const uint8_t *frames, uint32_t n, uint32_t w, uint32_t h, const char *dir, uint32_t first_index,
                        int threads, int level)
{
    if (!frames || !dir) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    if (n == 0) return D2R_OK;
    const size_t fb = (size_t)w * h * 3;
    D2rJobPool pool(std::min<int>((int)n, threads > 0 ? threads : d2r_default_io_threads()));
    const std::string d(dir);
    for (uint32_t i = 0; i < n; i++)
        pool.submit(0, [=](std::string &err) { return d2r_png_write_file(frames + fb * i, w, h, level, d2r_png_name(d, first_index + i), err); });
    pool.wait(-1);
    std::string err;
    int rc = pool.take_error(err);
    return rc ? d2r_fail(nullptr, rc, err) : D2R_OK;
}

int d2r_png_write_batch_bg(const uint8_t *frames, uint32_t n, uint32_t w, uint32_t h, const uint8_t *background, const char *dir,
                           uint32_t first_index, int threads)
{
    if (!frames || !dir || !background) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    if (n == 0) return D2R_OK;
    std::shared_ptr<const D2rPngBase> base = d2r_png_base_build(background, w, h);
    if (!base) return d2r_fail(nullptr, D2R_ERR_INVALID, "bad image size");
    const size_t fb = (size_t)w * h * 3;
    D2rJobPool pool(std::min<int>((int)n, threads > 0 ? threads : d2r_default_io_threads()));
    const std::string d(dir);
    for (uint32_t i = 0; i < n; i++)
        pool.submit(0, [=](std::string &err) { return d2r_png_write_file_delta(*base, frames + fb * i, d2r_png_name(d, first_index + i), err); });
    pool.wait(-1);
    std::string err;
    int rc = pool.take_error(err);
    return rc ? d2r_fail(nullptr, rc, err) : D2R_OK;
}

int d2r_png_read_batch(const char *dir, const uint32_t *indices, uint32_t first_index, uint32_t n, uint32_t w, uint32_t h,
                       uint8_t *frames_out, int threads)
{
    if (!dir || !frames_out) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    if (n == 0) return D2R_OK;
    if (w == 0 || h == 0) return d2r_fail(nullptr, D2R_ERR_INVALID, "bad image size");
    const size_t fb = (size_t)w * h * 3;
    D2rJobPool pool(std::min<int>((int)n, threads > 0 ? threads : d2r_default_io_threads()));
    const std::string d(dir);
    for (uint32_t i = 0; i < n; i++)
        pool.submit(0, [=](std::string &err) {
            return d2r_png_read_file(d2r_png_name(d, indices ? indices[i] : first_index + i), w, h, frames_out + fb * i, nullptr, nullptr, err);
        });
    pool.wait(-1);
    std::string err;
    int rc = pool.take_error(err);
    return rc ? d2r_fail(nullptr, rc, err) : D2R_OK;
}

int d2r_png_size(const char *path, uint32_t *w, uint32_t *h)
{
    if (!path || !w || !h) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    std::string err;
    int rc = d2r_png_read_file(path, 0, 0, nullptr, w, h, err);
    return rc ? d2r_fail(nullptr, rc, err) : D2R_OK;
}

}  // extern "C"
