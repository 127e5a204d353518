// ingp.hip — instant-ngp `.ingp` snapshot -> d2r_nerf, behind the C ABI (host code only).
//
// The reference loads `fg_base.ingp` / `bg_base.ingp` with pyngp.Testbed.load_snapshot (reference
// reconstruction/ngp_visual_model.py:24-28).  No snapshot and no instant-ngp source is available offline, so the
// format is read as believed (same reading as dream2real_amd/ingp.py, against
// which tests/test_ingp.py holds this loader): zlib/gzip-compressed msgpack of the network config plus
//   snapshot.params_binary        fp16: density MLP, colour MLP, then the hash-grid tables
//   snapshot.density_grid_binary  fp16 128^3 per cascade, Morton order
//   snapshot.nerf.dataset         per-image metadata, scale, offset, aabb_scale
// UNPINNED against real files: every constant comes from the snapshot, anything unexpected is an error.
#include <math.h>
#include <string.h>
#include <zlib.h>

#include <map>
#include <memory>

#include "d2r_internal.h"

namespace {

// ---- minimal msgpack DOM
struct Node {
    enum Kind { NIL, BOOL, INT, FLOAT, STR, BIN, ARR, MAP } kind = NIL;
    int64_t i = 0;
    double f = 0.0;
    const uint8_t *p = nullptr;      // STR / BIN payload (points into the decompressed buffer)
    size_t n = 0;
    std::vector<Node> arr;
    std::map<std::string, Node> map;

    const Node *get(const char *key) const
    {
        if (kind != MAP) return nullptr;
        auto it = map.find(key);
        return it == map.end() ? nullptr : &it->second;
    }
    bool number() const { return kind == INT || kind == FLOAT || kind == BOOL; }
    double num() const { return kind == FLOAT ? f : (double)i; }
    std::string str() const { return kind == STR ? std::string((const char *)p, n) : std::string(); }
};

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    int depth = 0;
    uint64_t be(int nbytes)
    {
        if ((size_t)(end - p) < (size_t)nbytes) { ok = false; return 0; }
        uint64_t v = 0;
        for (int k = 0; k < nbytes; k++) v = (v << 8) | *p++;
        return v;
    }
    bool take(size_t nbytes, const uint8_t **out)
    {
        if ((size_t)(end - p) < nbytes) { ok = false; return false; }
        *out = p;
        p += nbytes;
        return true;
    }
    Node parse()
    {
        Node nd;
        if (!ok || p >= end || ++depth > 64) { ok = false; return nd; }
        const uint8_t t = *p++;
        // a payload that runs past the end leaves an EMPTY node (never a null pointer with a length) and fails the parse
        auto str = [&](size_t len) { nd.kind = Node::STR; nd.n = ok && take(len, &nd.p) ? len : 0; };
        auto bin = [&](size_t len) { nd.kind = Node::BIN; nd.n = ok && take(len, &nd.p) ? len : 0; };
        auto arr = [&](size_t len) {
            nd.kind = Node::ARR;
            for (size_t k = 0; k < len && ok; k++) nd.arr.push_back(parse());
        };
        auto map = [&](size_t len) {
            nd.kind = Node::MAP;
            for (size_t k = 0; k < len && ok; k++) {
                Node key = parse();
                if (!ok) break;
                Node val = parse();
                if (!ok) break;
                nd.map[key.kind == Node::STR ? key.str() : ("#" + std::to_string(key.i))] = std::move(val);
            }
        };
        if (t <= 0x7f) { nd.kind = Node::INT; nd.i = t; }
        else if (t >= 0xe0) { nd.kind = Node::INT; nd.i = (int8_t)t; }
        else if ((t & 0xe0) == 0xa0) str(t & 0x1f);
        else if ((t & 0xf0) == 0x90) arr(t & 0x0f);
        else if ((t & 0xf0) == 0x80) map(t & 0x0f);
        else switch (t) {
            case 0xc0: break;
            case 0xc2: nd.kind = Node::BOOL; nd.i = 0; break;
            case 0xc3: nd.kind = Node::BOOL; nd.i = 1; break;
            case 0xc4: bin(be(1)); break;
            case 0xc5: bin(be(2)); break;
            case 0xc6: bin(be(4)); break;
            case 0xca: { uint32_t u = (uint32_t)be(4); float v; memcpy(&v, &u, 4); nd.kind = Node::FLOAT; nd.f = v; break; }
            case 0xcb: { uint64_t u = be(8); double v; memcpy(&v, &u, 8); nd.kind = Node::FLOAT; nd.f = v; break; }
            case 0xcc: nd.kind = Node::INT; nd.i = (int64_t)be(1); break;
            case 0xcd: nd.kind = Node::INT; nd.i = (int64_t)be(2); break;
            case 0xce: nd.kind = Node::INT; nd.i = (int64_t)be(4); break;
            case 0xcf: nd.kind = Node::INT; nd.i = (int64_t)be(8); break;
            case 0xd0: nd.kind = Node::INT; nd.i = (int8_t)be(1); break;
            case 0xd1: nd.kind = Node::INT; nd.i = (int16_t)be(2); break;
            case 0xd2: nd.kind = Node::INT; nd.i = (int32_t)be(4); break;
            case 0xd3: nd.kind = Node::INT; nd.i = (int64_t)be(8); break;
            case 0xd9: str(be(1)); break;
            case 0xda: str(be(2)); break;
            case 0xdb: str(be(4)); break;
            case 0xdc: arr(be(2)); break;
            case 0xdd: arr(be(4)); break;
            case 0xde: map(be(2)); break;
            case 0xdf: map(be(4)); break;
            default: ok = false;          // ext types do not occur in snapshots
        }
        depth--;
        return nd;
    }
};

// The largest snapshot this loader accepts once inflated: seven 128^3 fp16 density cascades are 29 MB and the tables of a
// 2^24-entry hash grid with F = 4 are 1 GiB, so 4 GiB is far beyond any model d2r_nerf_create would take, and it bounds
// what a crafted stream (a decompression bomb) can make the library allocate.
constexpr size_t kMaxInflated = (size_t)4 << 30;

// 0 = ok, 1 = malformed / truncated stream, 2 = inflated size over kMaxInflated
int inflate_all(const uint8_t *src, size_t len, std::vector<uint8_t> &out)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return 1;           // zlib or gzip header, auto-detected
    size_t fed = 0, produced = 0;                                // zlib's counters are 32-bit where uLong is: keep our own
    out.resize(std::min(kMaxInflated, std::max<size_t>(len * 4, 1 << 20)));
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        if (zs.avail_in == 0 && fed < len) {                     // input in pieces a uInt can count (files over 4 GiB)
            const size_t piece = std::min<size_t>(len - fed, 1u << 30);
            zs.next_in = (Bytef *)(src + fed);
            zs.avail_in = (uInt)piece;
            fed += piece;
        }
        if (produced == out.size()) {
            if (out.size() >= kMaxInflated) { inflateEnd(&zs); return 2; }
            out.resize(std::min(kMaxInflated, out.size() * 2));
        }
        const size_t room = std::min<size_t>(out.size() - produced, 1u << 30);
        zs.next_out = out.data() + produced;
        zs.avail_out = (uInt)room;
        rc = inflate(&zs, Z_NO_FLUSH);
        produced += room - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return 1; }
        if (rc == Z_OK && zs.avail_in == 0 && fed == len && zs.avail_out != 0) { inflateEnd(&zs); return 1; }   // truncated
    }
    out.resize(produced);
    inflateEnd(&zs);
    return 0;
}

float half_bits_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint32_t morton_compact(uint32_t v)
{
    v &= 0x09249249u;
    v = (v ^ (v >> 2)) & 0x030C30C3u;
    v = (v ^ (v >> 4)) & 0x0300F00Fu;
    v = (v ^ (v >> 8)) & 0xFF0000FFu;
    v = (v ^ (v >> 16)) & 0x000003FFu;
    return v;
}

}  // namespace

// Level table as tiny-cuda-nn's GridEncoding derives it: scale_l = exp2(l * log2(b)) * N_min - 1 in float,
// res = ceil(scale) + 1, sizes rounded up to 8 and capped at 2^log2_hashmap_size.  log2 / exp2 are evaluated in
// double and rounded to float (dream2real_amd/scene.py grid_levels does the same, so both loaders agree bit for bit).
void d2r_grid_levels(uint32_t L, uint32_t log2_hashmap, uint32_t base, double per_level_scale, uint32_t aabb_scale,
                     std::vector<float> &scale, std::vector<uint32_t> &res, std::vector<uint32_t> &size,
                     std::vector<uint32_t> &offset, uint32_t &n_entries)
{
    if (!(per_level_scale > 0.0)) per_level_scale = exp(log(2048.0 * aabb_scale / base) / (double)(L - 1));
    const float log2_pls = (float)log2((double)(float)per_level_scale);
    scale.resize(L); res.resize(L); size.resize(L); offset.resize(L);
    uint64_t off = 0;
    for (uint32_t l = 0; l < L; l++) {
        const float s = (float)exp2((double)((float)l * log2_pls)) * (float)base - 1.0f;
        const uint64_t r = (uint64_t)ceil((double)s) + 1;
        const uint64_t max_params = 0xffffffffull / 2;
        uint64_t p = (double)r * (double)r * (double)r > (double)max_params ? max_params : r * r * r;
        p = (p + 7) / 8 * 8;
        p = std::min<uint64_t>(p, 1ull << log2_hashmap);
        scale[l] = s; res[l] = (uint32_t)r; size[l] = (uint32_t)p; offset[l] = (uint32_t)off;
        off += p;
    }
    n_entries = (uint32_t)off;
}

static int load_ingp(d2r_ctx *ctx, const void *bytes, size_t len, d2r_nerf **out, d2r_ingp_info *info,
                     d2r_ingp_view *views, uint32_t views_cap)
{
    const uint8_t *src = (const uint8_t *)bytes;
    std::vector<uint8_t> raw;
    const bool gz = src[0] == 0x1f && src[1] == 0x8b, zl = src[0] == 0x78;
    if (gz || zl) {
        const int zrc = inflate_all(src, len, raw);
        if (zrc == 2) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: inflates to more than 4 GiB");
        if (zrc) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: decompression failed");
    } else {
        raw.assign(src, src + len);         // an uncompressed .msgpack
    }
    Reader rd{raw.data(), raw.data() + raw.size()};
    const Node cfg = rd.parse();
    if (!rd.ok || cfg.kind != Node::MAP) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: not a msgpack map");
    const Node *snap = cfg.get("snapshot"), *enc = cfg.get("encoding"), *net = cfg.get("network"), *rgb = cfg.get("rgb_network");
    if (!snap || !enc || !net || !rgb) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: encoding / network / rgb_network / snapshot missing");
    auto num = [](const Node *m, const char *k, double dflt) { const Node *v = m ? m->get(k) : nullptr; return v && v->number() ? v->num() : dflt; };
    auto text = [](const Node *m, const char *k, const char *dflt) { const Node *v = m ? m->get(k) : nullptr; return v && v->kind == Node::STR ? v->str() : std::string(dflt); };
    const std::string otype = text(enc, "otype", "HashGrid"), etype = text(enc, "type", "Hash");
    if ((otype != "HashGrid" && otype != "Grid") || etype != "Hash") return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: unsupported position encoding");
    if (num(net, "n_neurons", 64) != 64 || num(rgb, "n_neurons", 64) != 64 || num(net, "n_hidden_layers", 1) != 1 || num(rgb, "n_hidden_layers", 2) != 2)
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: only the 32->64->16 density and 32->64->64->16 colour MLPs are implemented");
    const Node *nerf = snap->get("nerf"), *ds = nerf ? nerf->get("dataset") : nullptr;
    const double aabb_d = num(nerf, "aabb_scale", num(ds, "aabb_scale", 1));
    const uint32_t aabb = aabb_d >= 1 && aabb_d <= 128 && aabb_d == floor(aabb_d) ? (uint32_t)aabb_d : 0;
    if (aabb == 0 || (aabb & (aabb - 1)) || aabb > 128) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: aabb_scale must be a power of two <= 128");
    uint32_t n_casc = 1;
    while ((1u << (n_casc - 1)) < aabb) n_casc++;
    const double L_d = num(enc, "n_levels", 16), F_d = num(enc, "n_features_per_level", 2);
    const uint32_t L = L_d >= 1 && L_d <= 64 ? (uint32_t)L_d : 0, F = F_d >= 1 && F_d <= 16 ? (uint32_t)F_d : 0;
    if (!((L == 16 && F == 2) || (L == 8 && F == 4))) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: hash grid layout must be L=16,F=2 or L=8,F=4");
    std::vector<float> scale;
    std::vector<uint32_t> res, size, offset;
    uint32_t n_entries = 0;
    // the level table is computed from these three: refuse values its arithmetic is not defined for (shift counts,
    // float -> integer conversions of inf / NaN) instead of deriving a table from them
    const double log2_hash = num(enc, "log2_hashmap_size", 19), base_res = num(enc, "base_resolution", 16),
                 pls = num(enc, "per_level_scale", 0.0);
    if (!(log2_hash >= 1 && log2_hash <= 24) || log2_hash != floor(log2_hash))
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: log2_hashmap_size must be an integer in 1..24");
    if (!(base_res >= 1 && base_res <= 65536) || base_res != floor(base_res))
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: base_resolution must be an integer in 1..65536");
    if (!(pls == 0.0 || (std::isfinite(pls) && pls >= 1.0 && pls <= 16.0)))      // 0 = absent: derived from aabb_scale
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: per_level_scale must be finite, in 1..16");
    d2r_grid_levels(L, (uint32_t)log2_hash, (uint32_t)base_res, pls, aabb, scale, res, size, offset, n_entries);
    if (text(snap, "params_type", "__half") != "__half") return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: params_type must be __half");
    const Node *pb = snap->get("params_binary"), *db = snap->get("density_grid_binary");
    if (!pb || pb->kind != Node::BIN || !db || db->kind != Node::BIN) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: params_binary / density_grid_binary missing");
    const size_t n_in = (size_t)L * F, sizes[6] = {64 * n_in, 16 * 64, 64 * 32, 64 * 64, 16 * 64, (size_t)n_entries * F};
    size_t total = 0;
    for (size_t s : sizes) total += s;
    if (pb->n != total * 2) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: params_binary holds " + std::to_string(pb->n / 2) + " halves, expected " + std::to_string(total));
    if (num(snap, "density_grid_size", 128) != (double)D2R_GRID) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: density_grid_size must be 128");
    const size_t cells = (size_t)D2R_GRID * D2R_GRID * D2R_GRID;
    if (db->n != (size_t)n_casc * cells * 2) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: density grid must hold 128^3 values per cascade");
    // params: copies (msgpack payloads are not aligned)
    std::vector<uint16_t> params(total);
    memcpy(params.data(), pb->p, pb->n);
    const uint16_t *part[6];
    {
        size_t o = 0;
        for (int k = 0; k < 6; k++) { part[k] = params.data() + o; o += sizes[k]; }
    }
    // occupancy: instant-ngp's update_density_grid_mean_and_bitfield (threshold min(0.01, mean of max(d, 0) over cascade
    // 0), Morton -> linear, coarser cascades OR in the 2x2x2 max-pool of the finer one) -- ingp.py occupancy_from_density
    std::vector<uint16_t> dens((size_t)n_casc * cells);
    memcpy(dens.data(), db->p, db->n);
    double mean = 0.0;
    for (size_t c = 0; c < cells; c++) mean += std::max(half_bits_to_float(dens[c]), 0.0f);
    mean /= (double)cells;
    const float thresh = (float)std::min(0.01, mean);
    std::vector<uint8_t> occ((size_t)n_casc * cells, 0);
    for (uint32_t cs = 0; cs < n_casc; cs++)
        for (uint32_t m = 0; m < cells; m++) {
            const uint32_t x = morton_compact(m), y = morton_compact(m >> 1), z = morton_compact(m >> 2);
            occ[(size_t)cs * cells + x + D2R_GRID * (y + D2R_GRID * z)] = half_bits_to_float(dens[(size_t)cs * cells + m]) > thresh;
        }
    for (uint32_t cs = 1; cs < n_casc; cs++)
        for (int z = 0; z < D2R_GRID; z++)
            for (int y = 0; y < D2R_GRID; y++)
                for (int x = 0; x < D2R_GRID; x++)
                    if (occ[(size_t)(cs - 1) * cells + x + D2R_GRID * (y + D2R_GRID * z)])
                        occ[(size_t)cs * cells + (32 + x / 2) + D2R_GRID * ((32 + y / 2) + D2R_GRID * (32 + z / 2))] = 1;
    std::vector<uint8_t> bits((size_t)n_casc * cells / 8, 0);
    for (size_t c = 0; c < (size_t)n_casc * cells; c++)
        if (occ[c]) bits[c >> 3] |= (uint8_t)(1u << (c & 7));
    d2r_nerf_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.n_levels = L; desc.n_features = F;
    desc.level_scale = scale.data(); desc.level_res = res.data(); desc.level_size = size.data(); desc.level_offset = offset.data();
    desc.n_entries = n_entries;
    desc.dw1_fp16 = part[0]; desc.dw2_fp16 = part[1]; desc.cw1_fp16 = part[2]; desc.cw2_fp16 = part[3]; desc.cw3_fp16 = part[4];
    desc.grid_fp16 = part[5];
    desc.occupancy_bits = bits.data();
    desc.aabb_scale = aabb;
    if (const Node *ra = snap->get("render_aabb")) {
        float v[6];
        bool have = false;
        if (ra->kind == Node::MAP && ra->get("min") && ra->get("max") && ra->get("min")->arr.size() == 3 && ra->get("max")->arr.size() == 3) {
            for (int k = 0; k < 3; k++) { v[k] = (float)ra->get("min")->arr[k].num(); v[3 + k] = (float)ra->get("max")->arr[k].num(); }
            have = true;
        } else if (ra->kind == Node::ARR && ra->arr.size() == 6) {
            for (int k = 0; k < 6; k++) v[k] = (float)ra->arr[k].num();
            have = true;
        }
        if (have) {
            const float half = 0.5f * (float)aabb;
            bool whole = true;
            for (int k = 0; k < 3; k++) whole = whole && v[k] <= 0.5f - half && v[3 + k] >= 0.5f + half;
            if (!whole) memcpy(desc.render_aabb, v, sizeof v);
        }
    }
    int rc = d2r_nerf_create(ctx, &desc, out);
    if (rc) return rc;
    if (info) {
        memset(info, 0, sizeof *info);
        info->n_levels = L; info->n_features = F; info->aabb_scale = aabb;
        info->dataset_scale = num(ds, "scale", 1.0);
        const Node *offn = ds ? ds->get("offset") : nullptr;
        for (int k = 0; k < 3; k++) info->dataset_offset[k] = offn && offn->arr.size() == 3 ? offn->arr[k].num() : 0.5;
        if (const Node *bc = snap->get("background_color"))
            if (bc->kind == Node::ARR && bc->arr.size() == 4) {
                info->has_background = 1;
                for (int k = 0; k < 4; k++) info->background_color[k] = (float)bc->arr[k].num();
            }
        const Node *md = ds ? ds->get("metadata") : nullptr;
        info->n_views = md && md->kind == Node::ARR ? (uint32_t)md->arr.size() : 0;
        for (uint32_t k = 0; k < info->n_views && k < views_cap && views; k++) {
            const Node &m = md->arr[k];
            const Node *r = m.get("resolution"), *fl = m.get("focal_length"), *pp = m.get("principal_point");
            if (!r || !fl || !pp || r->arr.size() != 2 || fl->arr.size() != 2 || pp->arr.size() != 2) continue;
            d2r_ingp_view &v = views[info->n_views_written++];
            const double rw = r->arr[0].num(), rh = r->arr[1].num();
            v.w = rw >= 0 && rw < 4294967296.0 ? (uint32_t)rw : 0; v.h = rh >= 0 && rh < 4294967296.0 ? (uint32_t)rh : 0;
            v.fx = fl->arr[0].num(); v.fy = fl->arr[1].num();
            v.cx = pp->arr[0].num() * v.w; v.cy = pp->arr[1].num() * v.h;
        }
    }
    return D2R_OK;
}

// ---- d2r_ingp_inspect: what is in a snapshot, and what of it the loader above reads (host only, no device)
namespace {

// every msgpack path load_ingp() looks at ("[]" = each element of an array of maps)
const char *const kPathsRead[] = {
    "encoding.otype", "encoding.type", "encoding.n_levels", "encoding.n_features_per_level", "encoding.log2_hashmap_size",
    "encoding.base_resolution", "encoding.per_level_scale", "network.n_neurons", "network.n_hidden_layers",
    "rgb_network.n_neurons", "rgb_network.n_hidden_layers", "snapshot.params_type", "snapshot.params_binary",
    "snapshot.density_grid_binary", "snapshot.density_grid_size", "snapshot.render_aabb", "snapshot.render_aabb.min",
    "snapshot.render_aabb.max", "snapshot.background_color", "snapshot.nerf.aabb_scale", "snapshot.nerf.dataset.aabb_scale",
    "snapshot.nerf.dataset.scale", "snapshot.nerf.dataset.offset", "snapshot.nerf.dataset.metadata[].resolution",
    "snapshot.nerf.dataset.metadata[].focal_length", "snapshot.nerf.dataset.metadata[].principal_point",
};

const char *kind_name(const Node &n)
{
    switch (n.kind) {
        case Node::NIL: return "nil";
        case Node::BOOL: return "bool";
        case Node::INT: return "int";
        case Node::FLOAT: return "float";
        case Node::STR: return "str";
        case Node::BIN: return "bin";
        case Node::ARR: return "array";
        default: return "map";
    }
}

std::string scalar_text(const Node &n)
{
    char buf[64];
    switch (n.kind) {
        case Node::BOOL: return n.i ? "true" : "false";
        case Node::INT: snprintf(buf, sizeof buf, "%lld", (long long)n.i); return buf;
        case Node::FLOAT: snprintf(buf, sizeof buf, "%.9g", n.f); return buf;
        case Node::STR: {
            std::string t = n.str().substr(0, 48);
            for (char &c : t) if ((unsigned char)c < 0x20 || c == 0x7f) c = '?';
            return "\"" + t + "\"";
        }
        default: return "";
    }
}

void walk(const Node &n, const std::string &path, std::string &out)
{
    if (n.kind == Node::MAP) {
        for (const auto &kv : n.map) walk(kv.second, path.empty() ? kv.first : path + "." + kv.first, out);
        return;
    }
    bool maps = n.kind == Node::ARR && !n.arr.empty();
    for (const Node &e : n.arr) maps = maps && e.kind == Node::MAP;
    if (maps) {                                    // array of maps (per-image metadata): the first element stands for all
        out += "- array " + std::to_string(n.arr.size()) + " " + path + "[]\n";
        walk(n.arr[0], path + "[]", out);
        return;
    }
    bool read = false;
    for (const char *r : kPathsRead) read = read || path == r;
    const size_t size = n.kind == Node::ARR ? n.arr.size() : (n.kind == Node::BIN || n.kind == Node::STR) ? n.n : 1;
    std::string value = scalar_text(n);
    if (n.kind == Node::ARR && n.arr.size() <= 16) {
        value = "[";
        for (size_t k = 0; k < n.arr.size(); k++) value += (k ? "," : "") + (n.arr[k].kind == Node::ARR ? std::string("[..]") : scalar_text(n.arr[k]));
        value += "]";
    }
    out += std::string(read ? "R " : "- ") + kind_name(n) + " " + std::to_string(size) + " " + path + (value.empty() ? "" : " = " + value) + "\n";
}

int inspect_ingp(const void *bytes, size_t len, std::string &text)
{
    const uint8_t *src = (const uint8_t *)bytes;
    std::vector<uint8_t> raw;
    if ((src[0] == 0x1f && src[1] == 0x8b) || src[0] == 0x78) {
        if (inflate_all(src, len, raw)) return d2r_fail(nullptr, D2R_ERR_INVALID, "snapshot: decompression failed");
    } else {
        raw.assign(src, src + len);
    }
    Reader rd{raw.data(), raw.data() + raw.size()};
    const Node cfg = rd.parse();
    if (!rd.ok || cfg.kind != Node::MAP) return d2r_fail(nullptr, D2R_ERR_INVALID, "snapshot: not a msgpack map");
    text = "# d2r_ingp_inspect: <R = read by d2r_nerf_load_ingp | - = ignored> <kind> <elements or bytes> <path> [= value]\n";
    text += "# inflated_bytes " + std::to_string(raw.size()) + " trailing_bytes " + std::to_string((size_t)(rd.end - rd.p)) + "\n";
    walk(cfg, "", text);
    // what the loader would derive, next to what the file holds
    auto num = [](const Node *m, const char *k, double dflt) { const Node *v = m ? m->get(k) : nullptr; return v && v->number() ? v->num() : dflt; };
    const Node *enc = cfg.get("encoding"), *snap = cfg.get("snapshot");
    const Node *nerf = snap ? snap->get("nerf") : nullptr, *ds = nerf ? nerf->get("dataset") : nullptr;
    const double L = num(enc, "n_levels", 16), F = num(enc, "n_features_per_level", 2), lh = num(enc, "log2_hashmap_size", 19),
                 br = num(enc, "base_resolution", 16), pls = num(enc, "per_level_scale", 0.0),
                 aabb = num(nerf, "aabb_scale", num(ds, "aabb_scale", 1));
    if (L >= 1 && L <= 64 && F >= 1 && F <= 16 && lh >= 1 && lh <= 24 && br >= 1 && br <= 65536 && aabb >= 1 && aabb <= 128 &&
        (pls == 0.0 || (std::isfinite(pls) && pls >= 1.0 && pls <= 16.0))) {
        std::vector<float> scale;
        std::vector<uint32_t> res, size, offset;
        uint32_t n_entries = 0;
        d2r_grid_levels((uint32_t)L, (uint32_t)lh, (uint32_t)br, pls, (uint32_t)aabb, scale, res, size, offset, n_entries);
        const size_t n_in = (size_t)(L * F), mlp = 64 * n_in + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64;
        const Node *pb = snap ? snap->get("params_binary") : nullptr, *db = snap ? snap->get("density_grid_binary") : nullptr;
        text += "# derived: grid_entries " + std::to_string(n_entries) + " grid_params " + std::to_string((size_t)n_entries * (size_t)F) +
                " mlp_params " + std::to_string(mlp) + " n_params_expected " + std::to_string((size_t)n_entries * (size_t)F + mlp) +
                " params_binary_halves " + std::to_string(pb && pb->kind == Node::BIN ? pb->n / 2 : 0) + "\n";
        uint32_t n_casc = 1;
        while ((1u << (n_casc - 1)) < (uint32_t)aabb) n_casc++;
        text += "# derived: cascades " + std::to_string(n_casc) + " density_grid_halves_expected " + std::to_string((size_t)n_casc * 128 * 128 * 128) +
                " density_grid_binary_halves " + std::to_string(db && db->kind == Node::BIN ? db->n / 2 : 0) + "\n";
        for (uint32_t l = 0; l < (uint32_t)L; l++) {
            char buf[128];
            snprintf(buf, sizeof buf, "# level %u: scale %.6f res %u entries %u offset %u\n", l, scale[l], res[l], size[l], offset[l]);
            text += buf;
        }
    } else {
        text += "# derived: (encoding fields outside the range the loader accepts)\n";
    }
    return D2R_OK;
}

}  // namespace

extern "C" int d2r_ingp_inspect(const void *bytes, size_t len, char *out, size_t cap, size_t *needed)
{
    if (!bytes || len < 4 || (!out && cap)) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    try {
        std::string text;
        const int rc = inspect_ingp(bytes, len, text);
        if (rc) return rc;
        if (needed) *needed = text.size() + 1;
        if (out && cap) {
            const size_t n = std::min(cap - 1, text.size());
            memcpy(out, text.data(), n);
            out[n] = 0;
        }
        return D2R_OK;
    } catch (const std::bad_alloc &) {
        return d2r_fail(nullptr, D2R_ERR_MEMORY, "snapshot: out of host memory");
    } catch (...) {
        return d2r_fail(nullptr, D2R_ERR_INVALID, "snapshot: malformed");
    }
}

// No C++ exception crosses the C ABI (include/d2r.h): allocation failures and anything a malformed snapshot provokes
// inside the standard library come back as error codes.
extern "C" int d2r_nerf_load_ingp(d2r_ctx *ctx, const void *bytes, size_t len, d2r_nerf **out, d2r_ingp_info *info,
                                  d2r_ingp_view *views, uint32_t views_cap)
{
    if (!ctx || !bytes || !out || len < 4) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    try {
        return load_ingp(ctx, bytes, len, out, info, views, views_cap);
    } catch (const std::bad_alloc &) {
        return d2r_fail(ctx, D2R_ERR_MEMORY, "snapshot: out of host memory");
    } catch (const std::exception &e) {
        return d2r_fail(ctx, D2R_ERR_INVALID, std::string("snapshot: malformed (") + e.what() + ")");
    } catch (...) {
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: malformed");
    }
}
