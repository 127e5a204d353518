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

const char *kind_name_of(const Node &n)
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

std::string scalar_text_of(const Node &n)
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

// ---- what of a snapshot defines the rendered function, and what this library implements of it.
// The layout is BELIEVED (no instant-ngp source or file offline): nothing is defaulted silently.  Keys that define the
// network or the frame are either READ (required, error naming the key when absent or of the wrong kind) or CHECKED
// (optional, but when present they must carry the one value the kernels implement); an unknown key inside a section
// that defines the function (encoding / network / rgb_network / dir_encoding) is an error; unknown keys elsewhere are
// counted (d2r_ingp_info.n_unknown_keys) and listed by d2r_ingp_inspect with a '?'.
namespace {

void walk(const Node &n, const std::string &path, std::string &out, uint32_t *unknown = nullptr);   // defined with d2r_ingp_inspect

struct Audit {
    std::string err;          // first failure ("" = none)
    uint32_t unknown = 0;     // keys that are neither read, checked nor known to be irrelevant to rendering
    bool fail(const std::string &m)
    {
        if (err.empty()) err = m;
        return false;
    }
};

const Node *need(Audit &A, const Node *m, const char *section, const char *key, int kind /* Node::Kind, or -1 = number */)
{
    const Node *v = m ? m->get(key) : nullptr;
    if (!v) {
        A.fail(std::string("snapshot: required key ") + section + "." + key + " is missing");
        return nullptr;
    }
    const bool ok = kind < 0 ? v->number() : (int)v->kind == kind;
    if (!ok) {
        A.fail(std::string("snapshot: ") + section + "." + key + " has msgpack kind '" + kind_name_of(*v) + "', expected " +
               (kind < 0 ? "a number" : kind == Node::STR ? "a string" : kind == Node::BIN ? "binary" : kind == Node::ARR ? "an array" : "a map"));
        return nullptr;
    }
    return v;
}

// optional key that must, when present, hold one of the accepted strings
void check_text(Audit &A, const Node *m, const char *section, const char *key, std::initializer_list<const char *> accepted)
{
    const Node *v = m ? m->get(key) : nullptr;
    if (!v) return;
    std::string got = v->kind == Node::STR ? v->str() : std::string("<") + kind_name_of(*v) + ">";
    for (const char *a : accepted)
        if (got == a) return;
    std::string list;
    for (const char *a : accepted) list += std::string(list.empty() ? "" : " / ") + a;
    A.fail(std::string("snapshot: ") + section + "." + key + " = '" + got + "' is not implemented (only " + list + ")");
}

// every scalar leaf of v (a number, a bool, or arrays of those to any depth: instant-ngp writes its ivec2 / vec3 fields as arrays) equals `want`
bool all_leaves_equal(const Node &v, double want, std::string &text)
{
    if (v.number()) {
        text += (text.empty() ? "" : ", ") + scalar_text_of(v);
        return v.num() == want;
    }
    if (v.kind != Node::ARR || v.arr.empty()) {
        text += std::string(text.empty() ? "" : ", ") + "<" + kind_name_of(v) + ">";
        return false;
    }
    bool ok = true;
    for (const Node &e : v.arr) ok = all_leaves_equal(e, want, text) && ok;
    return ok;
}

// optional key that must, when present, be a number equal to `want` — or a vector whose elements all are (upstream serialises e.g.
// NerfDataset::envmap_resolution as an ivec2, [0, 0] in every snapshot without an environment map)
void check_num(Audit &A, const Node *m, const char *section, const char *key, double want, const char *why)
{
    const Node *v = m ? m->get(key) : nullptr;
    if (!v) return;
    std::string text;
    if (!all_leaves_equal(*v, want, text))
        A.fail(std::string("snapshot: ") + section + "." + key + " = " + (v->kind == Node::ARR ? "[" + text + "]" : text) +
               " is not implemented (" + why + ")");
}

void only_keys(Audit &A, const Node *m, const char *section, std::initializer_list<const char *> known)
{
    if (!m || m->kind != Node::MAP) return;
    for (const auto &kv : m->map) {
        bool ok = false;
        for (const char *k : known) ok = ok || kv.first == k;
        if (!ok) A.fail(std::string("snapshot: unknown key ") + section + "." + kv.first + " in a section that defines the network: refusing to guess what it changes");
    }
}

bool is_identity3(const Node *m)
{
    if (!m || m->kind != Node::ARR) return false;
    std::vector<double> v;
    for (const Node &r : m->arr) {
        if (r.kind == Node::ARR) for (const Node &e : r.arr) v.push_back(e.num());
        else v.push_back(r.num());
    }
    if (v.size() != 9) return false;
    for (int k = 0; k < 9; k++)
        if (fabs(v[k] - (k % 4 == 0 ? 1.0 : 0.0)) > 1e-6) return false;
    return true;
}

// rendering-relevant state outside the network sections: must be absent or at the value that leaves the frame unchanged
void audit_render_state(Audit &A, const Node *snap, const Node *nerf, const Node *ds)
{
    for (const Node *m : {snap, ds})
        if (m && m->get("render_aabb_to_local") && !is_identity3(m->get("render_aabb_to_local")))
            A.fail(std::string("snapshot: ") + (m == snap ? "snapshot" : "snapshot.nerf.dataset") + ".render_aabb_to_local is not the identity: a rotated crop box is not implemented");
    check_num(A, snap, "snapshot", "exposure", 0.0, "frames are rendered at exposure 0");
    check_num(A, ds, "snapshot.nerf.dataset", "n_extra_learnable_dims", 0.0, "extra network input dimensions change the colour network's input width");
    check_num(A, nerf, "snapshot.nerf", "n_extra_dims", 0.0, "extra network input dimensions change the colour network's input width");
    check_num(A, ds, "snapshot.nerf.dataset", "envmap_resolution", 0.0, "environment maps are not implemented");
    for (const char *flag : {"is_hdr", "from_mitsuba"}) {
        const Node *v = ds ? ds->get(flag) : nullptr;
        if (v && v->number() && v->num() != 0.0)
            A.fail(std::string("snapshot: snapshot.nerf.dataset.") + flag + " is set: " +
                   (flag[0] == 'i' ? "HDR (linear-colour) datasets are not implemented" : "mitsuba camera conventions are not implemented"));
    }
    // tiny-cuda-nn activation enums as instant-ngp stores them (ENerfActivation: None 0, ReLU 1, Logistic 2, Exponential 3)
    check_num(A, nerf, "snapshot.nerf", "rgb_activation", 2.0, "colour = sigmoid(network output)");
    check_num(A, nerf, "snapshot.nerf", "density_activation", 3.0, "sigma = exp(network output)");
}

// metadata[k].lens as instant-ngp's to_json(Lens) writes it (believed): OpenCV = {k1, k2, p1, p2}; fisheye = {k1 .. k4}; f-theta
// = {ftheta_p0 ..}; {latlong: true}; {equirectangular: true}; a perspective lens writes no key (nil / an empty map).  The path
// renders every frame through set_camera_to_training_view's lens (reference reconstruction/combined_rendering.py:98,116), so the
// OpenCV form is READ (configs/shopping_demo.json:51-56 is what the reference's snapshots carry); the other models are not
// implemented and refuse the snapshot.  Returns false with `why` set on failure.
bool read_lens(const Node *ln, uint32_t &mode, float (&prm)[4], std::string &why)
{
    mode = D2R_LENS_PERSPECTIVE;
    for (float &p : prm) p = 0.f;
    if (!ln || ln->kind == Node::NIL || (ln->kind == Node::MAP && ln->map.empty())) return true;
    if (ln->kind != Node::MAP) { why = "is not a map"; return false; }
    auto on = [&](const char *k) { const Node *v = ln->get(k); return v && v->number() && v->num() != 0.0; };
    if (ln->get("k3") || ln->get("k4")) { why = "is an OpenCV fisheye lens (k3 / k4): not implemented"; return false; }
    for (const char *k : {"ftheta_p0", "ftheta_p1", "ftheta_p2", "ftheta_p3", "ftheta_p4", "w", "h"})
        if (ln->get(k)) { why = "is an f-theta lens: not implemented"; return false; }
    if (on("latlong") || on("equirectangular") || on("orthographic")) { why = "is a lat-long / equirectangular / orthographic lens: not implemented"; return false; }
    if (const Node *md = ln->get("mode")) {                // {mode, params} form: ELensMode Perspective 0, OpenCV 1
        if (!md->number() || (md->num() != 0.0 && md->num() != 1.0)) { why = "has a mode other than perspective (0) / OpenCV (1): not implemented"; return false; }
        const Node *pa = ln->get("params");
        if (md->num() == 1.0) {
            if (!pa || pa->kind != Node::ARR || pa->arr.size() < 4) { why = "lacks params (k1, k2, p1, p2)"; return false; }
            mode = D2R_LENS_OPENCV;
            for (int k = 0; k < 4; k++) prm[k] = (float)pa->arr[k].num();
        }
    } else {
        const char *keys[4] = {"k1", "k2", "p1", "p2"};
        int have = 0;
        for (const char *k : keys) have += ln->get(k) ? 1 : 0;
        if (have == 0) return true;                         // nothing lens-like: perspective
        if (have != 4) { why = "is an OpenCV lens without all of k1, k2, p1, p2"; return false; }
        mode = D2R_LENS_OPENCV;
        for (int k = 0; k < 4; k++) {
            const Node *v = ln->get(keys[k]);
            if (!v->number()) { why = std::string(keys[k]) + " is not a number"; return false; }
            prm[k] = (float)v->num();
        }
    }
    if (mode == D2R_LENS_OPENCV) {
        bool zero = true;
        for (float p : prm) {
            if (!std::isfinite(p) || fabsf(p) > 16.f) { why = "has a coefficient that is not finite or beyond 16"; return false; }
            zero = zero && p == 0.f;
        }
        if (zero) mode = D2R_LENS_PERSPECTIVE;              // an all-zero OpenCV lens is the pinhole: skip the iteration
    }
    return true;
}

void audit_networks(Audit &A, const Node *enc, const Node *net, const Node *rgb, const Node *dir)
{
    only_keys(A, enc, "encoding", {"otype", "type", "n_levels", "n_features_per_level", "log2_hashmap_size", "base_resolution",
                                   "per_level_scale", "interpolation", "n_dims_to_encode"});
    check_text(A, enc, "encoding", "otype", {"HashGrid", "Grid"});
    check_text(A, enc, "encoding", "type", {"Hash"});
    check_text(A, enc, "encoding", "interpolation", {"Linear"});
    check_num(A, enc, "encoding", "n_dims_to_encode", 3.0, "positions are 3-D");
    for (const Node *m : {net, rgb}) {
        const char *sec = m == net ? "network" : "rgb_network";
        only_keys(A, m, sec, {"otype", "activation", "output_activation", "n_neurons", "n_hidden_layers"});
        check_text(A, m, sec, "otype", {"FullyFusedMLP", "CutlassMLP"});
        check_text(A, m, sec, "activation", {"ReLU"});
        check_text(A, m, sec, "output_activation", {"None"});
    }
    // direction encoding: spherical harmonics of degree 4 on the 3 direction inputs, alone or as the first member of a
    // Composite whose other members are Identity encodings of the (zero) extra dimensions
    auto sh4 = [&](const Node *m, const char *sec) {
        only_keys(A, m, sec, {"otype", "degree", "n_dims_to_encode"});
        check_num(A, m, sec, "n_dims_to_encode", 3.0, "directions are 3-D");
        const Node *dg = need(A, m, sec, "degree", -1);
        if (dg && dg->num() != 4.0) A.fail(std::string("snapshot: ") + sec + ".degree = " + scalar_text_of(*dg) + " is not implemented (spherical harmonics of degree 4 only)");
    };
    const Node *ot = need(A, dir, "dir_encoding", "otype", Node::STR);
    if (!ot) return;
    if (ot->str() == "SphericalHarmonics") sh4(dir, "dir_encoding");
    else if (ot->str() == "Composite") {
        only_keys(A, dir, "dir_encoding", {"otype", "nested"});
        const Node *ns = need(A, dir, "dir_encoding", "nested", Node::ARR);
        if (!ns) return;
        if (ns->arr.empty() || ns->arr[0].kind != Node::MAP || !ns->arr[0].get("otype") || ns->arr[0].get("otype")->str() != "SphericalHarmonics")
            A.fail("snapshot: dir_encoding.nested[0] must be the SphericalHarmonics encoding of the direction");
        else sh4(&ns->arr[0], "dir_encoding.nested[0]");
        for (size_t k = 1; k < ns->arr.size(); k++) {
            const Node *o = ns->arr[k].kind == Node::MAP ? ns->arr[k].get("otype") : nullptr;
            if (!o || o->str() != "Identity")
                A.fail("snapshot: dir_encoding.nested[" + std::to_string(k) + "] is not an Identity encoding of the (absent) extra dimensions: not implemented");
        }
    } else {
        A.fail("snapshot: dir_encoding.otype = '" + ot->str() + "' is not implemented (SphericalHarmonics, or Composite of it)");
    }
}

}  // namespace

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
    Audit A;
    const Node *snap = need(A, &cfg, "<root>", "snapshot", Node::MAP), *enc = need(A, &cfg, "<root>", "encoding", Node::MAP),
               *net = need(A, &cfg, "<root>", "network", Node::MAP), *rgb = need(A, &cfg, "<root>", "rgb_network", Node::MAP),
               *dir = need(A, &cfg, "<root>", "dir_encoding", Node::MAP);
    if (!A.err.empty()) return d2r_fail(ctx, D2R_ERR_INVALID, A.err);
    auto num = [](const Node *m, const char *k, double dflt) { const Node *v = m ? m->get(k) : nullptr; return v && v->number() ? v->num() : dflt; };
    auto req = [&](const Node *m, const char *sec, const char *k) { const Node *v = need(A, m, sec, k, -1); return v ? v->num() : 0.0; };
    audit_networks(A, enc, net, rgb, dir);
    const double nn = req(net, "network", "n_neurons"), nh = req(net, "network", "n_hidden_layers"),
                 rn = req(rgb, "rgb_network", "n_neurons"), rh = req(rgb, "rgb_network", "n_hidden_layers");
    const double L_d = req(enc, "encoding", "n_levels"), F_d = req(enc, "encoding", "n_features_per_level"),
                 log2_hash = req(enc, "encoding", "log2_hashmap_size"), base_res = req(enc, "encoding", "base_resolution"),
                 pls = num(enc, "per_level_scale", 0.0);        // optional: instant-ngp derives it from aabb_scale when absent
    const Node *nerf = need(A, snap, "snapshot", "nerf", Node::MAP), *ds = need(A, nerf, "snapshot.nerf", "dataset", Node::MAP);
    if (!A.err.empty()) return d2r_fail(ctx, D2R_ERR_INVALID, A.err);
    audit_render_state(A, snap, nerf, ds);
    if (!A.err.empty()) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, A.err);
    if (nn != 64 || rn != 64 || nh != 1 || rh != 2)
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: only the 32->64->16 density and 32->64->64->16 colour MLPs are implemented (network.n_neurons " +
                        std::to_string((long long)nn) + ", n_hidden_layers " + std::to_string((long long)nh) + "; rgb_network.n_neurons " + std::to_string((long long)rn) +
                        ", n_hidden_layers " + std::to_string((long long)rh) + ")");
    // aabb_scale lives on the NeRF state and on its dataset; at least one must be there, and they must agree
    const Node *a1 = nerf->get("aabb_scale"), *a2 = ds->get("aabb_scale");
    if (!a1 && !a2) return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: required key snapshot.nerf.aabb_scale (or snapshot.nerf.dataset.aabb_scale) is missing");
    if (a1 && a2 && a1->num() != a2->num())
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: snapshot.nerf.aabb_scale and snapshot.nerf.dataset.aabb_scale disagree");
    const double aabb_d = (a1 ? a1 : a2)->number() ? (a1 ? a1 : a2)->num() : 0.0;
    const uint32_t aabb = aabb_d >= 1 && aabb_d <= 128 && aabb_d == floor(aabb_d) ? (uint32_t)aabb_d : 0;
    if (aabb == 0 || (aabb & (aabb - 1)) || aabb > 128) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: aabb_scale must be a power of two <= 128");
    uint32_t n_casc = 1;
    while ((1u << (n_casc - 1)) < aabb) n_casc++;
    // the marcher's step rule is tied to the box: constant steps for aabb_scale 1, cone angle 1/256 beyond it
    check_num(A, nerf, "snapshot.nerf", "cone_angle_constant", aabb <= 1 ? 0.0 : 1.0 / 256.0,
              aabb <= 1 ? "aabb_scale 1 marches with constant steps" : "aabb_scale >= 2 marches with cone angle 1/256");
    if (!A.err.empty()) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, A.err);
    const uint32_t L = L_d >= 1 && L_d <= 64 ? (uint32_t)L_d : 0, F = F_d >= 1 && F_d <= 16 ? (uint32_t)F_d : 0;
    if (!((L == 16 && F == 2) || (L == 8 && F == 4)))
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: hash grid layout must be L=16,F=2 or L=8,F=4 (encoding.n_levels " + std::to_string((long long)L_d) +
                        ", n_features_per_level " + std::to_string((long long)F_d) + ")");
    std::vector<float> scale;
    std::vector<uint32_t> res, size, offset;
    uint32_t n_entries = 0;
    // the level table is computed from these three: refuse values its arithmetic is not defined for (shift counts,
    // float -> integer conversions of inf / NaN) instead of deriving a table from them
    if (!(log2_hash >= 1 && log2_hash <= 24) || log2_hash != floor(log2_hash))
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: log2_hashmap_size must be an integer in 1..24");
    if (!(base_res >= 1 && base_res <= 65536) || base_res != floor(base_res))
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: base_resolution must be an integer in 1..65536");
    if (!(pls == 0.0 || (std::isfinite(pls) && pls >= 1.0 && pls <= 16.0)))      // 0 = absent: derived from aabb_scale
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: per_level_scale must be finite, in 1..16");
    d2r_grid_levels(L, (uint32_t)log2_hash, (uint32_t)base_res, pls, aabb, scale, res, size, offset, n_entries);
    const Node *pt = need(A, snap, "snapshot", "params_type", Node::STR);
    const Node *pb = need(A, snap, "snapshot", "params_binary", Node::BIN), *db = need(A, snap, "snapshot", "density_grid_binary", Node::BIN);
    const double dgs = req(snap, "snapshot", "density_grid_size");
    if (!A.err.empty()) return d2r_fail(ctx, D2R_ERR_INVALID, A.err);
    const size_t n_in = (size_t)L * F, sizes[6] = {64 * n_in, 16 * 64, 64 * 32, 64 * 64, 16 * 64, (size_t)n_entries * F};
    size_t total = 0;
    for (size_t s : sizes) total += s;
    if (pt->str() != "__half")
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: snapshot.params_type = '" + pt->str() + "': only __half (fp16) parameters are implemented" +
                        (pb->n == total * 4 ? " (params_binary has the size of fp32 parameters)" : ""));
    if (const Node *np = snap->get("n_params"))
        if (np->number() && np->num() != (double)total)
            return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: snapshot.n_params = " + scalar_text_of(*np) + ", but the encoding / network sections describe " +
                            std::to_string(total) + " parameters (density MLP, colour MLP, hash tables): the parameter layout is not the one this loader knows");
    if (pb->n != total * 2)
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: snapshot.params_binary holds " + std::to_string(pb->n / 2) + " halves, expected " + std::to_string(total) +
                        " (density MLP " + std::to_string(sizes[0] + sizes[1]) + " + colour MLP " + std::to_string(sizes[2] + sizes[3] + sizes[4]) + " + hash tables " +
                        std::to_string(sizes[5]) + ")" + (pb->n == total * 4 ? ": the size of fp32 parameters" : ""));
    if (dgs != (double)D2R_GRID) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: snapshot.density_grid_size must be 128");
    const size_t cells = (size_t)D2R_GRID * D2R_GRID * D2R_GRID;
    if (db->n != (size_t)n_casc * cells * 2)
        return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: snapshot.density_grid_binary holds " + std::to_string(db->n) + " bytes, expected fp16 x 128^3 x " +
                        std::to_string(n_casc) + " cascade(s) = " + std::to_string((size_t)n_casc * cells * 2) +
                        (db->n == (size_t)n_casc * cells * 4 ? " (it has the size of fp32 densities)" : db->n == (size_t)n_casc * cells / 8 ? " (it has the size of a bitfield)" : ""));
    // what nerf_matrix_to_ngp needs of the dataset
    const double ds_scale = req(ds, "snapshot.nerf.dataset", "scale");
    const Node *offn = need(A, ds, "snapshot.nerf.dataset", "offset", Node::ARR);
    if (A.err.empty() && offn->arr.size() != 3) A.fail("snapshot: snapshot.nerf.dataset.offset must hold 3 numbers");
    if (!A.err.empty()) return d2r_fail(ctx, D2R_ERR_INVALID, A.err);
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
    // ctx == nullptr: d2r_ingp_validate — everything above ran (every check a load makes on the file), nothing is created
    int rc = ctx ? d2r_nerf_create(ctx, &desc, out) : D2R_OK;
    if (rc) return rc;
    if (info) {
        memset(info, 0, sizeof *info);
        {
            std::string listing;
            walk(cfg, "", listing, &info->n_unknown_keys);
        }
        info->n_levels = L; info->n_features = F; info->aabb_scale = aabb;
        info->dataset_scale = ds_scale;
        for (int k = 0; k < 3; k++) info->dataset_offset[k] = offn->arr[k].num();
        if (const Node *bc = snap->get("background_color"))
            if (bc->kind == Node::ARR && bc->arr.size() == 4) {
                info->has_background = 1;
                for (int k = 0; k < 4; k++) info->background_color[k] = (float)bc->arr[k].num();
            }
        const Node *md = ds ? ds->get("metadata") : nullptr;
        info->n_views = md && md->kind == Node::ARR ? (uint32_t)md->arr.size() : 0;
        for (uint32_t k = 0; k < info->n_views && (ctx ? (k < views_cap && views) : true); k++) {
            const Node &m = md->arr[k];
            const Node *r = m.get("resolution"), *fl = m.get("focal_length"), *pp = m.get("principal_point");
            if (!r || !fl || !pp || r->arr.size() != 2 || fl->arr.size() != 2 || pp->arr.size() != 2) {
                if (ctx) {
                    d2r_nerf_destroy(*out);
                    *out = nullptr;
                }
                return d2r_fail(ctx, D2R_ERR_INVALID, "snapshot: snapshot.nerf.dataset.metadata[" + std::to_string(k) + "] lacks " +
                                (!r || r->arr.size() != 2 ? "resolution" : !fl || fl->arr.size() != 2 ? "focal_length" : "principal_point") +
                                " (2 numbers): set_camera_to_training_view cannot take its intrinsics from it");
            }
            d2r_ingp_view scratch;
            d2r_ingp_view &v = (views && k < views_cap) ? views[info->n_views_written++] : scratch;
            const double rw = r->arr[0].num(), rh = r->arr[1].num();
            v.w = rw >= 0 && rw < 4294967296.0 ? (uint32_t)rw : 0; v.h = rh >= 0 && rh < 4294967296.0 ? (uint32_t)rh : 0;
            v.fx = fl->arr[0].num(); v.fy = fl->arr[1].num();
            v.cx = pp->arr[0].num() * v.w; v.cy = pp->arr[1].num() * v.h;
            std::string why;
            if (!read_lens(m.get("lens"), v.lens_mode, v.lens_params, why)) {
                if (ctx) {
                    d2r_nerf_destroy(*out);
                    *out = nullptr;
                }
                return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "snapshot: snapshot.nerf.dataset.metadata[" + std::to_string(k) + "].lens " + why +
                                " (set_camera_to_training_view makes this lens the render lens)");
            }
        }
        if (const Node *rl = nerf->get("render_with_lens_distortion")) info->render_with_lens_distortion = rl->number() && rl->num() != 0.0 ? 1 : 0;
    }
    return D2R_OK;
}

// ---- d2r_ingp_inspect: what is in a snapshot, and what of it the loader above reads (host only, no device)
namespace {

// every msgpack path load_ingp() reads ("[]" = each element of an array of maps)
const char *const kPathsRead[] = {
    "encoding.n_levels", "encoding.n_features_per_level", "encoding.log2_hashmap_size",
    "encoding.base_resolution", "encoding.per_level_scale", "network.n_neurons", "network.n_hidden_layers",
    "rgb_network.n_neurons", "rgb_network.n_hidden_layers", "snapshot.params_type", "snapshot.params_binary",
    "snapshot.density_grid_binary", "snapshot.density_grid_size", "snapshot.render_aabb", "snapshot.render_aabb.min",
    "snapshot.render_aabb.max", "snapshot.background_color", "snapshot.nerf.aabb_scale", "snapshot.nerf.dataset.aabb_scale",
    "snapshot.nerf.dataset.scale", "snapshot.nerf.dataset.offset", "snapshot.nerf.dataset.metadata[].resolution",
    "snapshot.nerf.dataset.metadata[].focal_length", "snapshot.nerf.dataset.metadata[].principal_point",
    "snapshot.nerf.dataset.metadata[].lens", "snapshot.nerf.dataset.metadata[].lens.k1", "snapshot.nerf.dataset.metadata[].lens.k2",
    "snapshot.nerf.dataset.metadata[].lens.p1", "snapshot.nerf.dataset.metadata[].lens.p2", "snapshot.nerf.dataset.metadata[].lens.mode",
    "snapshot.nerf.dataset.metadata[].lens.params", "snapshot.nerf.render_with_lens_distortion",
};
// paths the loader CHECKS: they change the rendered function, and only one value of each is implemented (an error otherwise)
const char *const kPathsChecked[] = {
    "encoding.otype", "encoding.type", "encoding.interpolation", "encoding.n_dims_to_encode",
    "network.otype", "network.activation", "network.output_activation",
    "rgb_network.otype", "rgb_network.activation", "rgb_network.output_activation",
    "dir_encoding.otype", "dir_encoding.degree", "dir_encoding.n_dims_to_encode", "dir_encoding.nested[].otype",
    "dir_encoding.nested[].degree", "dir_encoding.nested[].n_dims_to_encode",
    "snapshot.n_params", "snapshot.exposure", "snapshot.render_aabb_to_local", "snapshot.nerf.dataset.render_aabb_to_local",
    "snapshot.nerf.dataset.n_extra_learnable_dims", "snapshot.nerf.n_extra_dims", "snapshot.nerf.dataset.envmap_resolution",
    "snapshot.nerf.dataset.is_hdr", "snapshot.nerf.dataset.from_mitsuba", "snapshot.nerf.rgb_activation",
    "snapshot.nerf.density_activation", "snapshot.nerf.cone_angle_constant",
    // lens models other than perspective / OpenCV refuse the snapshot (read_lens)
    "snapshot.nerf.dataset.metadata[].lens.k3", "snapshot.nerf.dataset.metadata[].lens.k4", "snapshot.nerf.dataset.metadata[].lens.latlong",
    "snapshot.nerf.dataset.metadata[].lens.equirectangular", "snapshot.nerf.dataset.metadata[].lens.orthographic",
};
// prefixes known NOT to change what d2r renders through the path's entry points: training state, optimiser / loss
// configuration, the GUI's own camera and lights, bookkeeping.  (dataset.up / up_dir rotate the GUI's orbit and the
// poses instant-ngp READS from a transforms file; the path hands camera matrices to set_nerf_camera_matrix, which
// applies scale / offset / axis cycle only.)
const char *const kPrefixesIrrelevant[] = {
    "loss", "optimizer", "envmap", "distortion_map", "parent", "snapshot.version", "snapshot.mode", "snapshot.camera", "snapshot.up_dir",
    "snapshot.sun_dir", "snapshot.aabb", "snapshot.bounding_radius", "snapshot.training_step", "snapshot.loss",
    "snapshot.density_grid_ema_step", "snapshot.nerf.dataset.paths", "snapshot.nerf.dataset.xforms", "snapshot.nerf.dataset.n_images",
    "snapshot.nerf.dataset.up", "snapshot.nerf.dataset.wants_importance_sampling", "snapshot.nerf.dataset.has_rays",
    "snapshot.nerf.dataset.metadata[].rolling_shutter", "snapshot.nerf.dataset.metadata[].light_dir",
    "snapshot.nerf.dataset.metadata[].depth_scale", "snapshot.nerf.dataset.render_aabb", "snapshot.nerf.cam_", "snapshot.nerf.extra_dims_opt",
    "snapshot.nerf.rgb", "snapshot.nerf.training", "snapshot.nerf.sharpen", "snapshot.nerf.density_grid", "dir_encoding.nested[].n_bins",
    "dir_encoding.nested", "snapshot.nerf.dataset.metadata",
};

// 'R' read, 'C' checked, '-' known not to affect rendering, '?' unknown
char classify(const std::string &path)
{
    for (const char *r : kPathsRead)
        if (path == r) return 'R';
    for (const char *r : kPathsChecked)
        if (path == r) return 'C';
    for (const char *r : kPrefixesIrrelevant) {
        const size_t n = strlen(r);
        if (path.compare(0, n, r) == 0 && (path.size() == n || path[n] == '.' || path[n] == '[' || r[n - 1] == '_')) return '-';
    }
    return '?';
}

void walk(const Node &n, const std::string &path, std::string &out, uint32_t *unknown)
{
    if (n.kind == Node::MAP) {
        for (const auto &kv : n.map) walk(kv.second, path.empty() ? kv.first : path + "." + kv.first, out, unknown);
        return;
    }
    bool maps = n.kind == Node::ARR && !n.arr.empty();
    for (const Node &e : n.arr) maps = maps && e.kind == Node::MAP;
    if (maps) {                                    // array of maps (per-image metadata): the first element stands for all
        out += "- array " + std::to_string(n.arr.size()) + " " + path + "[]\n";
        walk(n.arr[0], path + "[]", out, unknown);
        return;
    }
    const char cls = classify(path);
    if (cls == '?' && unknown) ++*unknown;
    const size_t size = n.kind == Node::ARR ? n.arr.size() : (n.kind == Node::BIN || n.kind == Node::STR) ? n.n : 1;
    std::string value = scalar_text_of(n);
    if (n.kind == Node::ARR && n.arr.size() <= 16) {
        value = "[";
        for (size_t k = 0; k < n.arr.size(); k++) value += (k ? "," : "") + (n.arr[k].kind == Node::ARR ? std::string("[..]") : scalar_text_of(n.arr[k]));
        value += "]";
    }
    out += std::string(1, cls) + " " + kind_name_of(n) + " " + std::to_string(size) + " " + path + (value.empty() ? "" : " = " + value) + "\n";
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
    text = "# d2r_ingp_inspect: <R = read by d2r_nerf_load_ingp | C = checked: changes the rendered function, one value implemented | - = known not to "
           "affect rendering | ? = unknown to the loader> <kind> <elements or bytes> <path> [= value]\n";
    text += "# inflated_bytes " + std::to_string(raw.size()) + " trailing_bytes " + std::to_string((size_t)(rd.end - rd.p)) + "\n";
    uint32_t unknown = 0;
    walk(cfg, "", text, &unknown);
    text += "# unknown_keys " + std::to_string(unknown) + " (keys marked '?': not known to the loader, not known to be harmless)\n";
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

// Host only: would d2r_nerf_load_ingp take these bytes?  Runs every check the loader makes on the file (layout, kinds,
// sizes, the values of keys that change the rendered function) without a device; the message of a refusal names the key.
extern "C" int d2r_ingp_validate(const void *bytes, size_t len, d2r_ingp_info *info)
{
    if (!bytes || len < 4) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    try {
        d2r_ingp_info local;
        return load_ingp(nullptr, bytes, len, nullptr, info ? info : &local, nullptr, 0);
    } catch (const std::bad_alloc &) {
        return d2r_fail(nullptr, D2R_ERR_MEMORY, "snapshot: out of host memory");
    } catch (const std::exception &e) {
        return d2r_fail(nullptr, D2R_ERR_INVALID, std::string("snapshot: malformed (") + e.what() + ")");
    } catch (...) {
        return d2r_fail(nullptr, D2R_ERR_INVALID, "snapshot: malformed");
    }
}
