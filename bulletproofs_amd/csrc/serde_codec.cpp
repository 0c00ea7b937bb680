// Wire formats either side of the path: the serde encodings of the reference's proof structs
//   RangeProof           /root/reference/src/proofs/range_proof.rs:32-42   (+ InnerProductArg, inner_product.rs:27-33)
//   RangeProofWIP        /root/reference/src/proofs/range_proof_wip.rs:93-97 (+ WeightedInnerProdArg, weighted_inner_product.rs:35-44)
// as `#[derive(Serialize, Deserialize)]` produces them with curv-kzen 0.10's impls for Point / Scalar / BigInt, converted
// to and from this library's canonical proof bytes (include/bpmsm.h).  Host code only: no GPU call, no context.
//
// curv-kzen 0.10 is not under /root/reference and the reference never serialises anything (no test, no bench), so the
// three leaf encodings are a restatement of curv's published serde impls - PARITY UNPINNED:
//   Point<E>   struct { curve: "secp256k1", point: bytes }   bytes = to_bytes(true), 33-byte SEC1 compressed
//   Scalar<E>  struct { curve: "secp256k1", scalar: bytes }  bytes = 32-byte big-endian
//   BigInt     human-readable formats: hex string of to_bytes() (minimal big-endian magnitude, zero = "00");
//              binary formats: serialize_bytes(to_bytes())
// Each lives in ONE function below (put_point / put_scalar / put_bigint and their readers).  Formats:
//   BP_SERDE_JSON     serde_json::to_string: field names, declaration order, byte strings as arrays of numbers
//   BP_SERDE_BINCODE  bincode 1.x default options: fields in order, u64 little-endian lengths for str / bytes / Vec
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bpmsm.h"

namespace bp {
void set_error_msg(const char* msg);  // capi.cpp (the sanitizer harness serde_codec_san.cpp supplies its own)
}

namespace {

const char kCurve[] = "secp256k1";

struct Shape {
    int kind;    // 0 RangeProof, 1 RangeProofWIP
    size_t k;    // rounds
};
// canonical layout: BP = A S T1 T2 | tau_x miu tx | L[k] | R[k] | a b (292 + 66 k); BP+ = A | L[k] | R[k] | a_tag b_tag | r' s' d' (195 + 66 k)
bool shape_of(int kind, size_t len, Shape& s) {
    const size_t fixed = kind == 0 ? 292 : 195;
    if ((kind != 0 && kind != 1) || len < fixed || (len - fixed) % 66) return false;
    s.kind = kind;
    s.k = (len - fixed) / 66;
    return s.k <= 64;
}
bool sec1_ok(const uint8_t* p) { return p[0] == 2 || p[0] == 3; }

// ---------------------------------------------------------------------------------------------- writers
struct JsonW {
    std::string o;
    void bytes(const uint8_t* p, size_t n) {
        o += '[';
        for (size_t i = 0; i < n; i++) {
            if (i) o += ',';
            o += std::to_string((unsigned)p[i]);
        }
        o += ']';
    }
    void put_point(const uint8_t* sec1) {
        o += "{\"curve\":\"";
        o += kCurve;
        o += "\",\"point\":";
        bytes(sec1, 33);
        o += '}';
    }
    void put_scalar(const uint8_t* be32) {
        o += "{\"curve\":\"";
        o += kCurve;
        o += "\",\"scalar\":";
        bytes(be32, 32);
        o += '}';
    }
    void put_bigint(const uint8_t* be32) {
        size_t z = 0;
        while (z < 31 && be32[z] == 0) z++;  // minimal magnitude, zero keeps one byte
        static const char hx[] = "0123456789abcdef";
        o += '"';
        for (size_t i = z; i < 32; i++) {
            o += hx[be32[i] >> 4];
            o += hx[be32[i] & 15];
        }
        o += '"';
    }
    void key(const char* k, bool first = false) {
        if (!first) o += ',';
        o += '"';
        o += k;
        o += "\":";
    }
    void points(const uint8_t* p, size_t n) {
        o += '[';
        for (size_t i = 0; i < n; i++) {
            if (i) o += ',';
            put_point(p + 33 * i);
        }
        o += ']';
    }
};

struct BinW {
    std::vector<uint8_t> o;
    void u64(uint64_t v) {
        for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i)));
    }
    void blob(const void* p, size_t n) {
        u64(n);
        o.insert(o.end(), (const uint8_t*)p, (const uint8_t*)p + n);
    }
    void put_point(const uint8_t* sec1) {
        blob(kCurve, 9);
        blob(sec1, 33);
    }
    void put_scalar(const uint8_t* be32) {
        blob(kCurve, 9);
        blob(be32, 32);
    }
    void put_bigint(const uint8_t* be32) {
        size_t z = 0;
        while (z < 31 && be32[z] == 0) z++;
        blob(be32 + z, 32 - z);
    }
    void points(const uint8_t* p, size_t n) {
        u64(n);
        for (size_t i = 0; i < n; i++) put_point(p + 33 * i);
    }
};

// ---------------------------------------------------------------------------------------------- JSON reader (DOM)
struct JVal {
    enum { Null, Num, Str, Arr, Obj, Bool } t = Null;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const {
        if (t != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
struct JsonR {
    const char *p, *e;
    int depth = 0;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool lit(const char* s) {
        size_t n = strlen(s);
        if ((size_t)(e - p) < n || memcmp(p, s, n)) return false;
        p += n;
        return true;
    }
    bool str(std::string& out) {
        if (p >= e || *p != '"') return false;
        p++;
        while (p < e && *p != '"') {
            if (*p == '\\') {  // field names, curve name and hex digits need no escapes: accept the simple ones only
                if (p + 1 >= e) return false;
                char c = p[1];
                if (c == '"' || c == '\\' || c == '/')
                    out += c;
                else
                    return false;
                p += 2;
            } else {
                out += *p++;
            }
        }
        if (p >= e) return false;
        p++;
        return true;
    }
    bool val(JVal& v) {
        if (++depth > 16) return false;
        ws();
        if (p >= e) return false;
        bool ok = false;
        if (*p == '{') {
            p++;
            v.t = JVal::Obj;
            ws();
            if (p < e && *p == '}') {
                p++;
                ok = true;
            } else {
                for (;;) {
                    ws();
                    std::string k;
                    if (!str(k)) break;
                    ws();
                    if (p >= e || *p != ':') break;
                    p++;
                    JVal c;
                    if (!val(c)) break;
                    v.obj.emplace_back(std::move(k), std::move(c));
                    ws();
                    if (p < e && *p == ',') {
                        p++;
                        continue;
                    }
                    if (p < e && *p == '}') {
                        p++;
                        ok = true;
                    }
                    break;
                }
            }
        } else if (*p == '[') {
            p++;
            v.t = JVal::Arr;
            ws();
            if (p < e && *p == ']') {
                p++;
                ok = true;
            } else {
                for (;;) {
                    JVal c;
                    if (!val(c)) break;
                    v.arr.push_back(std::move(c));
                    ws();
                    if (p < e && *p == ',') {
                        p++;
                        continue;
                    }
                    if (p < e && *p == ']') {
                        p++;
                        ok = true;
                    }
                    break;
                }
            }
        } else if (*p == '"') {
            v.t = JVal::Str;
            ok = str(v.str);
        } else if (lit("true") || lit("false")) {
            v.t = JVal::Bool;
            ok = true;
        } else if (lit("null")) {
            v.t = JVal::Null;
            ok = true;
        } else {
            const char* s = p;
            while (p < e && ((*p >= '0' && *p <= '9') || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) p++;
            if (p > s && p - s < 32) {
                char buf[32];
                memcpy(buf, s, p - s);
                buf[p - s] = 0;
                char* end = nullptr;
                v.num = strtod(buf, &end);
                v.t = JVal::Num;
                ok = end && *end == 0;
            }
        }
        depth--;
        return ok;
    }
};

bool j_bytes(const JVal* v, size_t want, uint8_t* out) {
    if (!v || v->t != JVal::Arr || v->arr.size() != want) return false;
    for (size_t i = 0; i < want; i++) {
        const JVal& c = v->arr[i];
        if (c.t != JVal::Num || c.num < 0 || c.num > 255 || c.num != (double)(int)c.num) return false;
        out[i] = (uint8_t)(int)c.num;
    }
    return true;
}
bool j_curve(const JVal* v) {
    const JVal* c = v ? v->get("curve") : nullptr;
    return c && c->t == JVal::Str && c->str == kCurve;
}
bool get_point(const JVal* v, uint8_t* sec1) { return j_curve(v) && j_bytes(v->get("point"), 33, sec1) && sec1_ok(sec1); }
bool get_scalar(const JVal* v, uint8_t* be32) { return j_curve(v) && j_bytes(v->get("scalar"), 32, be32); }
bool get_bigint(const JVal* v, uint8_t* be32) {
    if (!v || v->t != JVal::Str) return false;
    const std::string& h = v->str;
    if (h.empty() || h.size() > 64) return false;  // canonical proofs hold 256-bit values
    memset(be32, 0, 32);
    const size_t n = h.size();
    for (size_t i = 0; i < n; i++) {
        const char c = h[n - 1 - i];
        int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
        if (d < 0) return false;
        be32[31 - i / 2] |= (uint8_t)(d << (4 * (i & 1)));
    }
    return true;
}
bool get_points(const JVal* v, std::vector<uint8_t>& out) {
    if (!v || v->t != JVal::Arr || v->arr.size() > 64) return false;
    out.resize(33 * v->arr.size());
    for (size_t i = 0; i < v->arr.size(); i++)
        if (!get_point(&v->arr[i], out.data() + 33 * i)) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------- bincode reader
struct BinR {
    const uint8_t *p, *e;
    bool u64(uint64_t& v) {
        if (e - p < 8) return false;
        v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
        p += 8;
        return true;
    }
    bool blob(const uint8_t*& b, size_t& n, size_t maxn) {
        uint64_t l;
        if (!u64(l) || l > maxn || (uint64_t)(e - p) < l) return false;
        b = p;
        n = (size_t)l;
        p += n;
        return true;
    }
    bool curve() {
        const uint8_t* b;
        size_t n;
        return blob(b, n, 32) && n == 9 && !memcmp(b, kCurve, 9);
    }
    bool get_point(uint8_t* sec1) {
        const uint8_t* b;
        size_t n;
        if (!curve() || !blob(b, n, 65) || n != 33) return false;
        memcpy(sec1, b, 33);
        return sec1_ok(sec1);
    }
    bool get_scalar(uint8_t* be32) {
        const uint8_t* b;
        size_t n;
        if (!curve() || !blob(b, n, 32) || n != 32) return false;
        memcpy(be32, b, 32);
        return true;
    }
    bool get_bigint(uint8_t* be32) {
        const uint8_t* b;
        size_t n;
        if (!blob(b, n, 32)) return false;
        memset(be32, 0, 32);
        memcpy(be32 + 32 - n, b, n);
        return true;
    }
    bool get_points(std::vector<uint8_t>& out) {
        uint64_t n;
        if (!u64(n) || n > 64) return false;
        out.resize(33 * (size_t)n);
        for (size_t i = 0; i < (size_t)n; i++)
            if (!get_point(out.data() + 33 * i)) return false;
        return true;
    }
};

template <class W>
void write_fields(W& w, const Shape& s, const uint8_t* c);

template <>
void write_fields<JsonW>(JsonW& w, const Shape& s, const uint8_t* c) {
    const size_t k = s.k;
    w.o += '{';
    if (s.kind == 0) {
        static const char* pn[4] = {"A", "S", "T1", "T2"};
        static const char* sn[3] = {"tau_x", "miu", "tx"};
        for (int i = 0; i < 4; i++) {
            w.key(pn[i], i == 0);
            w.put_point(c + 33 * i);
        }
        for (int i = 0; i < 3; i++) {
            w.key(sn[i]);
            w.put_scalar(c + 132 + 32 * i);
        }
        const uint8_t* L = c + 228;
        w.key("inner_product_proof");
        w.o += '{';
        w.key("L", true);
        w.points(L, k);
        w.key("R");
        w.points(L + 33 * k, k);
        w.key("a_tag");
        w.put_bigint(L + 66 * k);
        w.key("b_tag");
        w.put_bigint(L + 66 * k + 32);
        w.o += '}';
    } else {
        w.key("A", true);
        w.put_point(c);
        const uint8_t* L = c + 33;
        w.key("weighted_inner_product_proof");
        w.o += '{';
        w.key("L", true);
        w.points(L, k);
        w.key("R");
        w.points(L + 33 * k, k);
        const uint8_t* t = L + 66 * k;
        w.key("a_tag");
        w.put_point(t);
        w.key("b_tag");
        w.put_point(t + 33);
        w.key("r_prime");
        w.put_bigint(t + 66);
        w.key("s_prime");
        w.put_bigint(t + 98);
        w.key("delta_prime");
        w.put_bigint(t + 130);
        w.o += '}';
    }
    w.o += '}';
}

template <>
void write_fields<BinW>(BinW& w, const Shape& s, const uint8_t* c) {
    const size_t k = s.k;
    if (s.kind == 0) {
        for (int i = 0; i < 4; i++) w.put_point(c + 33 * i);
        for (int i = 0; i < 3; i++) w.put_scalar(c + 132 + 32 * i);
        const uint8_t* L = c + 228;
        w.points(L, k);
        w.points(L + 33 * k, k);
        w.put_bigint(L + 66 * k);
        w.put_bigint(L + 66 * k + 32);
    } else {
        w.put_point(c);
        const uint8_t* L = c + 33;
        w.points(L, k);
        w.points(L + 33 * k, k);
        const uint8_t* t = L + 66 * k;
        w.put_point(t);
        w.put_point(t + 33);
        w.put_bigint(t + 66);
        w.put_bigint(t + 98);
        w.put_bigint(t + 130);
    }
}

int fail(const char* msg) {
    bp::set_error_msg(msg);
    return BP_EINVAL;
}

}  // namespace

extern "C" {

int bp_proof_to_serde(int kind, int format, const uint8_t* proof, size_t proof_len, uint8_t* out, size_t cap, size_t* out_len) {
    Shape s;
    if (!proof || !out_len || !shape_of(kind, proof_len, s)) return fail("proof_to_serde: not a canonical proof length for this kind");
    // points must carry a SEC1 compressed prefix (the identity has no curv encoding either)
    const size_t npts_head = kind == 0 ? 4 : 1;
    for (size_t i = 0; i < npts_head; i++)
        if (!sec1_ok(proof + 33 * i)) return fail("proof_to_serde: bad SEC1 prefix");
    const uint8_t* L = proof + (kind == 0 ? 228 : 33);
    const size_t nlr = 2 * s.k + (kind == 1 ? 2 : 0);
    for (size_t i = 0; i < nlr; i++)
        if (!sec1_ok(L + 33 * i)) return fail("proof_to_serde: bad SEC1 prefix");
    const uint8_t* data;
    size_t n;
    JsonW jw;
    BinW bw;
    if (format == BP_SERDE_JSON) {
        write_fields(jw, s, proof);
        data = (const uint8_t*)jw.o.data();
        n = jw.o.size();
    } else if (format == BP_SERDE_BINCODE) {
        write_fields(bw, s, proof);
        data = bw.o.data();
        n = bw.o.size();
    } else {
        return fail("proof_to_serde: unknown format");
    }
    *out_len = n;  // the size needed is reported even when the buffer is too small
    if (!out || cap < n) return fail("proof_to_serde: output buffer too small (out_len holds the size needed)");
    memcpy(out, data, n);
    return BP_OK;
}

int bp_proof_from_serde(int kind, int format, const uint8_t* in, size_t in_len, uint8_t* proof_out, size_t cap, size_t* proof_len) {
    if (!in || !proof_len || (kind != 0 && kind != 1)) return fail("proof_from_serde: bad arguments");
    uint8_t head[4 * 33 + 3 * 32], tail[2 * 33 + 3 * 32];
    std::vector<uint8_t> L, R;
    bool ok = false;
    if (format == BP_SERDE_JSON) {
        JsonR r{(const char*)in, (const char*)in + in_len};
        JVal root;
        if (!r.val(root)) return fail("proof_from_serde: not JSON");
        r.ws();
        if (r.p != r.e || root.t != JVal::Obj) return fail("proof_from_serde: trailing characters / not an object");
        if (kind == 0) {
            const JVal* ip = root.get("inner_product_proof");
            ok = get_point(root.get("A"), head) && get_point(root.get("S"), head + 33) && get_point(root.get("T1"), head + 66) &&
                 get_point(root.get("T2"), head + 99) && get_scalar(root.get("tau_x"), head + 132) && get_scalar(root.get("miu"), head + 164) &&
                 get_scalar(root.get("tx"), head + 196) && ip && get_points(ip->get("L"), L) && get_points(ip->get("R"), R) &&
                 get_bigint(ip->get("a_tag"), tail) && get_bigint(ip->get("b_tag"), tail + 32);
        } else {
            const JVal* wp = root.get("weighted_inner_product_proof");
            ok = get_point(root.get("A"), head) && wp && get_points(wp->get("L"), L) && get_points(wp->get("R"), R) &&
                 get_point(wp->get("a_tag"), tail) && get_point(wp->get("b_tag"), tail + 33) && get_bigint(wp->get("r_prime"), tail + 66) &&
                 get_bigint(wp->get("s_prime"), tail + 98) && get_bigint(wp->get("delta_prime"), tail + 130);
        }
    } else if (format == BP_SERDE_BINCODE) {
        BinR r{in, in + in_len};
        if (kind == 0) {
            ok = r.get_point(head) && r.get_point(head + 33) && r.get_point(head + 66) && r.get_point(head + 99) && r.get_scalar(head + 132) &&
                 r.get_scalar(head + 164) && r.get_scalar(head + 196) && r.get_points(L) && r.get_points(R) && r.get_bigint(tail) &&
                 r.get_bigint(tail + 32);
        } else {
            ok = r.get_point(head) && r.get_points(L) && r.get_points(R) && r.get_point(tail) && r.get_point(tail + 33) &&
                 r.get_bigint(tail + 66) && r.get_bigint(tail + 98) && r.get_bigint(tail + 130);
        }
        ok = ok && r.p == r.e;
    } else {
        return fail("proof_from_serde: unknown format");
    }
    if (!ok) return fail("proof_from_serde: field missing, wrong curve, wrong length or bad SEC1 prefix");
    if (L.size() != R.size()) return fail("proof_from_serde: L and R differ in length");
    const size_t nhead = kind == 0 ? 228 : 33, ntail = kind == 0 ? 64 : 162;
    const size_t n = nhead + L.size() + R.size() + ntail;
    *proof_len = n;
    if (!proof_out || cap < n) return fail("proof_from_serde: output buffer too small (proof_len holds the size needed)");
    memcpy(proof_out, head, nhead);
    if (!L.empty()) {  // k = 0 (nm = 1) has no rounds
        memcpy(proof_out + nhead, L.data(), L.size());
        memcpy(proof_out + nhead + L.size(), R.data(), R.size());
    }
    memcpy(proof_out + nhead + L.size() + R.size(), tail, ntail);
    return BP_OK;
}

}  // extern "C"
