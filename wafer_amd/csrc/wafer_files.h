// Array / scalar files of the host driver in the five formats Wafer reads and writes
// (output.rs:85-400, input.rs:60-720): Messagepack, Csv, Json, Yaml, Ron.
//
//   * Csv: one `i,j,k,data` record per work-area cell, C order, no header (output.rs:148-165).
//   * the other four: ndarray 0.11's serde layout -- a struct {v: 1u8, dim: [nx, ny, nz],
//     data: [..]} -- through rmp-serde 0.13 (structs as ARRAYS, no field names), serde_json
//     (pretty), serde_yaml 0.7, ron (Cargo.toml:29-45).  None of those crates is in the
//     reference tree, so the byte-level layouts below restate their published formats; the
//     readers accept both the struct-as-array and struct-as-map msgpack forms and any
//     whitespace / pretty-printing of the text forms.
//   * hazard SURVEY.md 8a #8: the reference's non-CSV readers compare the file's unpadded dims
//     with the padded target and therefore always interpolate; the CSV branch's logic
//     (input.rs:640-656: same size -> embed in the zero frame, else trilinear resample) is
//     applied to every format here.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <sys/stat.h>
#include <vector>

enum WaferFileType { WF_MPK = 0, WF_CSV = 1, WF_JSON = 2, WF_YAML = 3, WF_RON = 4 };
static const char *kFileTypes[] = {"Messagepack", "Csv", "Json", "Yaml", "Ron"};
static const char *kFileExt[] = {".mpk", ".csv", ".json", ".yaml", ".ron"};

// an unpadded [nx][ny][nz] array, or (potential_sub only) a single value
struct FieldFile {
    uint32_t nx = 0, ny = 0, nz = 0;
    std::vector<double> data;
    bool scalar = false;
    double value = 0.0;
};

// ---------------------------------------------------------------------------
// numbers as text
// ---------------------------------------------------------------------------
// shortest round-trip digits d1 d2 ... dn and the decimal exponent of d1
static void shortest_digits(double av, std::string &digits, int &exp10)
{
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, av, std::chars_format::scientific);
    std::string s(buf, r.ptr); // d.ddde[+-]xx
    const size_t e = s.find('e');
    exp10 = atoi(s.c_str() + e + 1);
    digits.clear();
    for (size_t i = 0; i < e; ++i)
        if (s[i] != '.') digits += s[i];
}

// positional or exponent notation chosen by the decimal exponent, always with a fractional part
// ("1.0", "0.001", "1e-7", "1.5e21"): ryu (serde_json, csv) switches below 1e-5 and from 1e16,
// dtoa (serde_yaml 0.7) below 1e-6 and from 1e21
static std::string float_text(double v, int lo_exp, int hi_exp, const char *nan, const char *pinf, const char *ninf)
{
    if (std::isnan(v)) return nan;
    if (std::isinf(v)) return v < 0 ? ninf : pinf;
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    std::string d;
    int e;
    shortest_digits(std::fabs(v), d, e);
    std::string out = v < 0 ? "-" : "";
    if (e < lo_exp || e >= hi_exp) {
        out += d.substr(0, 1);
        if (d.size() > 1) out += "." + d.substr(1);
        out += "e" + std::to_string(e);
    } else if (e < 0) {
        out += "0." + std::string((size_t)(-e - 1), '0') + d;
    } else if ((size_t)e + 1 >= d.size()) {
        out += d + std::string((size_t)e + 1 - d.size(), '0') + ".0";
    } else {
        out += d.substr(0, (size_t)e + 1) + "." + d.substr((size_t)e + 1);
    }
    return out;
}
static std::string num_text(double v) { return float_text(v, -5, 16, "NaN", "inf", "-inf"); }        // csv, json
static std::string yaml_num(double v) { return float_text(v, -6, 21, ".nan", ".inf", "-.inf"); }
// Rust's `{}` on f64 (ron): shortest digits, never an exponent, no ".0" on integral values
static std::string rust_display(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[512];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

// ---------------------------------------------------------------------------
// messagepack (rmp-serde 0.13 `Serializer::new`: compact, structs as arrays)
// ---------------------------------------------------------------------------
struct MpkOut {
    std::vector<unsigned char> b;
    void byte(unsigned v) { b.push_back((unsigned char)v); }
    void be(uint64_t v, int n)
    {
        for (int i = n - 1; i >= 0; --i) byte((unsigned)(v >> (8 * i)) & 0xffu);
    }
    void array_len(uint64_t n)
    {
        if (n < 16) byte(0x90u | (unsigned)n);
        else if (n < 65536) { byte(0xdc); be(n, 2); }
        else { byte(0xdd); be(n, 4); }
    }
    void uint(uint64_t v) // the smallest encoding, as rmp's write_uint
    {
        if (v < 128) byte((unsigned)v);
        else if (v < 256) { byte(0xcc); be(v, 1); }
        else if (v < 65536) { byte(0xcd); be(v, 2); }
        else if (v < (1ull << 32)) { byte(0xce); be(v, 4); }
        else { byte(0xcf); be(v, 8); }
    }
    void f64(double v)
    {
        uint64_t u;
        memcpy(&u, &v, 8);
        byte(0xcb);
        be(u, 8);
    }
    bool save(const std::string &path) const
    {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
        return fclose(f) == 0 && ok;
    }
};

struct MpkIn {
    const unsigned char *p, *end;
    bool ok = true;
    explicit MpkIn(const std::vector<unsigned char> &v) : p(v.data()), end(v.data() + v.size()) {}
    unsigned byte()
    {
        if (p >= end) { ok = false; return 0; }
        return *p++;
    }
    uint64_t be(int n)
    {
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 8) | byte();
        return v;
    }
    unsigned peek() const { return p < end ? *p : 0xc1u; }
    // array or map header; is_map tells which
    bool container(uint64_t &n, bool &is_map)
    {
        const unsigned t = byte();
        is_map = false;
        if ((t & 0xf0u) == 0x90u) { n = t & 0x0fu; return ok; }
        if (t == 0xdc) { n = be(2); return ok; }
        if (t == 0xdd) { n = be(4); return ok; }
        is_map = true;
        if ((t & 0xf0u) == 0x80u) { n = t & 0x0fu; return ok; }
        if (t == 0xde) { n = be(2); return ok; }
        if (t == 0xdf) { n = be(4); return ok; }
        return ok = false;
    }
    bool str(std::string &s)
    {
        const unsigned t = byte();
        uint64_t n;
        if ((t & 0xe0u) == 0xa0u) n = t & 0x1fu;
        else if (t == 0xd9) n = be(1);
        else if (t == 0xda) n = be(2);
        else if (t == 0xdb) n = be(4);
        else return ok = false;
        if ((uint64_t)(end - p) < n) return ok = false;
        s.assign((const char *)p, (size_t)n);
        p += n;
        return ok;
    }
    bool number(double &v)
    {
        const unsigned t = byte();
        if (t < 0x80u) { v = t; return ok; }
        if (t >= 0xe0u) { v = (int8_t)t; return ok; }
        switch (t) {
        case 0xcc: v = (double)be(1); return ok;
        case 0xcd: v = (double)be(2); return ok;
        case 0xce: v = (double)be(4); return ok;
        case 0xcf: v = (double)be(8); return ok;
        case 0xd0: v = (int8_t)be(1); return ok;
        case 0xd1: v = (int16_t)be(2); return ok;
        case 0xd2: v = (int32_t)be(4); return ok;
        case 0xd3: v = (double)(int64_t)be(8); return ok;
        case 0xca: { uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); v = f; return ok; }
        case 0xcb: { uint64_t u = be(8); memcpy(&v, &u, 8); return ok; }
        default: return ok = false;
        }
    }
};

static bool slurp(const std::string &path, std::vector<unsigned char> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

static bool file_exists(const std::string &path)
{
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// ---------------------------------------------------------------------------
// writers
// ---------------------------------------------------------------------------
// `at(i, j, k)` yields the value of work cell (i, j, k)
template <typename At>
static bool write_field(const std::string &path, int file_type, uint32_t nx, uint32_t ny, uint32_t nz, At at, std::string &err)
{
    if (file_type == WF_MPK) { // [1, [nx, ny, nz], [data...]]
        MpkOut m;
        m.b.reserve((size_t)nx * ny * nz * 9 + 32);
        m.array_len(3);
        m.uint(1);
        m.array_len(3);
        m.uint(nx); m.uint(ny); m.uint(nz);
        m.array_len((uint64_t)nx * ny * nz);
        for (uint32_t i = 0; i < nx; ++i)
            for (uint32_t j = 0; j < ny; ++j)
                for (uint32_t k = 0; k < nz; ++k) m.f64(at(i, j, k));
        if (!m.save(path)) { err = "CreateFile: " + path; return false; }
        return true;
    }
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { err = "CreateFile: " + path; return false; }
    if (file_type == WF_CSV) {
        for (uint32_t i = 0; i < nx; ++i)
            for (uint32_t j = 0; j < ny; ++j)
                for (uint32_t k = 0; k < nz; ++k) fprintf(f, "%u,%u,%u,%s\n", i, j, k, num_text(at(i, j, k)).c_str());
    } else if (file_type == WF_JSON) { // serde_json::to_writer_pretty: two-space indent, one element per line
        fprintf(f, "{\n  \"v\": 1,\n  \"dim\": [\n    %u,\n    %u,\n    %u\n  ],\n  \"data\": [", nx, ny, nz);
        bool first = true;
        for (uint32_t i = 0; i < nx; ++i)
            for (uint32_t j = 0; j < ny; ++j)
                for (uint32_t k = 0; k < nz; ++k) {
                    fprintf(f, "%s\n    %s", first ? "" : ",", num_text(at(i, j, k)).c_str());
                    first = false;
                }
        fprintf(f, "%s]\n}", first ? "" : "\n  ");
    } else if (file_type == WF_YAML) {
        fprintf(f, "---\nv: 1\ndim:\n  - %u\n  - %u\n  - %u\ndata:\n", nx, ny, nz);
        for (uint32_t i = 0; i < nx; ++i)
            for (uint32_t j = 0; j < ny; ++j)
                for (uint32_t k = 0; k < nz; ++k) fprintf(f, "  - %s\n", yaml_num(at(i, j, k)).c_str());
    } else { // ron, PrettyConfig::default(): four-space indent, trailing commas
        fprintf(f, "(\n    v: 1,\n    dim: (%u, %u, %u,),\n    data: [\n", nx, ny, nz);
        for (uint32_t i = 0; i < nx; ++i)
            for (uint32_t j = 0; j < ny; ++j)
                for (uint32_t k = 0; k < nz; ++k) fprintf(f, "        %s,\n", rust_display(at(i, j, k)).c_str());
        fprintf(f, "    ],\n)");
    }
    if (fclose(f) != 0) { err = "Flush: " + path; return false; }
    return true;
}

// the work area of a padded [x][y][z] host array (output.rs:85-97, 379-400)
static bool write_array(const std::string &path, int file_type, const double *padded, uint32_t nx, uint32_t ny,
                        uint32_t nz, uint32_t e, std::string &err)
{
    const size_t py = ny + 2 * e, pz = nz + 2 * e;
    return write_field(path, file_type, nx, ny, nz,
                       [&](uint32_t i, uint32_t j, uint32_t k) { return padded[((size_t)(i + e) * py + (j + e)) * pz + (k + e)]; }, err);
}

// numpy .npy (version 1.0), C order float64, shape (nx + 2 pad, ny + 2 pad, nz + 2 pad): the work
// area surrounded by `pad` zero cells.  Not one of the reference's formats: the hand-over to the
// multi-rank driver, which memory-maps it (wafer_amd/run.py).
static bool write_npy(const std::string &path, const FieldFile &a, uint32_t pad, std::string &err)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "CreateFile: " + path; return false; }
    const size_t px = a.nx + 2 * (size_t)pad, py = a.ny + 2 * (size_t)pad, pz = a.nz + 2 * (size_t)pad;
    std::string head = "{'descr': '<f8', 'fortran_order': False, 'shape': (" +
                       (a.scalar ? std::string() : std::to_string(px) + ", " + std::to_string(py) + ", " + std::to_string(pz)) + "), }";
    while ((10 + head.size() + 1) % 64 != 0) head.push_back(' ');
    head.push_back('\n');
    const unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(head.size() & 0xff),
                                     (unsigned char)(head.size() >> 8)};
    bool ok = fwrite(magic, 1, 10, f) == 10 && fwrite(head.data(), 1, head.size(), f) == head.size();
    if (a.scalar) ok = ok && fwrite(&a.value, sizeof(double), 1, f) == 1; // PotentialSubSingle: a 0-d array
    const std::vector<double> zeros(pz, 0.0);
    std::vector<double> row(pz, 0.0);
    for (size_t i = 0; ok && !a.scalar && i < px; ++i)
        for (size_t j = 0; ok && j < py; ++j) {
            const bool inside = i >= pad && i < pad + a.nx && j >= pad && j < pad + a.ny;
            if (inside) memcpy(&row[pad], &a.data[((i - pad) * a.ny + (j - pad)) * (size_t)a.nz], sizeof(double) * a.nz);
            ok = fwrite(inside ? row.data() : zeros.data(), sizeof(double), pz, f) == pz;
        }
    if (fclose(f) != 0) ok = false;
    if (!ok) err = "CreateFile: short write to " + path;
    return ok;
}

// PotentialSubSingle { pot_sub } (output.rs:224-377)
static bool write_scalar_sub(const std::string &path, int file_type, double v, std::string &err)
{
    if (file_type == WF_MPK) {
        MpkOut m;
        m.array_len(1);
        m.f64(v);
        if (!m.save(path)) { err = "CreateFile: " + path; return false; }
        return true;
    }
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { err = "CreateFile: " + path; return false; }
    switch (file_type) {
    case WF_CSV: fprintf(f, "%s\n", rust_display(v).c_str()); break; // write_record(&[value.to_string()])
    case WF_JSON: fprintf(f, "{\n  \"pot_sub\": %s\n}", num_text(v).c_str()); break;
    case WF_YAML: fprintf(f, "---\npot_sub: %s\n", yaml_num(v).c_str()); break;
    default: fprintf(f, "(\n    pot_sub: %s,\n)", rust_display(v).c_str()); break;
    }
    fclose(f);
    return true;
}

// ---------------------------------------------------------------------------
// readers
// ---------------------------------------------------------------------------
static bool finish_shape(FieldFile &out, const std::string &path, std::string &err)
{
    if (out.data.size() != (size_t)out.nx * out.ny * out.nz || out.data.empty()) { // Array3::from_shape_vec: ErrorKind::ArrayShape
        err = "ArrayShape: " + path + ": " + std::to_string(out.data.size()) + " values do not fill " +
              std::to_string(out.nx) + "x" + std::to_string(out.ny) + "x" + std::to_string(out.nz);
        return false;
    }
    return true;
}

// input.rs:607-662: `i,j,k,data` records, dims = max index + 1; a file holding one bare value is
// a singular potential_sub (input.rs:340-375)
static bool read_csv_field(const std::string &path, FieldFile &out, std::string &err)
{
    std::ifstream f(path);
    if (!f) { err = "FileNotFound: " + path; return false; }
    unsigned mi = 0, mj = 0, mk = 0;
    std::string line;
    bool first = true;
    while (std::getline(f, line)) {
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
        unsigned i, j, k;
        double d;
        if (sscanf(line.c_str(), "%u,%u,%u,%lf", &i, &j, &k, &d) != 4) {
            char *endp = nullptr;
            const double v = strtod(line.c_str(), &endp);
            if (first && endp != line.c_str() && line.find(',') == std::string::npos) {
                out.scalar = true;
                out.value = v;
                return true;
            }
            err = "ParsePlainRecord: " + path;
            return false;
        }
        first = false;
        if (i > mi) mi = i;
        if (j > mj) mj = j;
        if (k > mk) mk = k;
        out.data.push_back(d);
    }
    out.nx = mi + 1; out.ny = mj + 1; out.nz = mk + 1;
    return finish_shape(out, path, err);
}

static bool read_mpk_field(const std::string &path, FieldFile &out, std::string &err)
{
    std::vector<unsigned char> raw;
    if (!slurp(path, raw)) { err = "FileNotFound: " + path; return false; }
    MpkIn m(raw);
    uint64_t n;
    bool is_map;
    if (!m.container(n, is_map)) { err = "Deserialize: " + path; return false; }
    if (n == 1) { // PotentialSubSingle
        std::string key;
        if (is_map && !m.str(key)) { err = "Deserialize: " + path; return false; }
        out.scalar = true;
        if (!m.number(out.value)) { err = "Deserialize: " + path; return false; }
        return true;
    }
    if (n != 3) { err = "Deserialize: " + path + ": not an ndarray {v, dim, data}"; return false; }
    for (int field = 0; field < 3; ++field) {
        std::string key = field == 0 ? "v" : field == 1 ? "dim" : "data";
        if (is_map && !m.str(key)) break;
        if (key == "v") {
            double v;
            if (!m.number(v) || v != 1.0) { err = "Deserialize: " + path + ": unknown ndarray format version"; return false; }
        } else if (key == "dim") {
            uint64_t nd;
            bool mm;
            double d[3];
            if (!m.container(nd, mm) || mm || nd != 3 || !m.number(d[0]) || !m.number(d[1]) || !m.number(d[2])) break;
            out.nx = (uint32_t)d[0]; out.ny = (uint32_t)d[1]; out.nz = (uint32_t)d[2];
        } else if (key == "data") {
            uint64_t len;
            bool mm;
            if (!m.container(len, mm) || mm) break;
            out.data.resize(len);
            for (uint64_t i = 0; i < len && m.ok; ++i) m.number(out.data[i]);
        } else {
            m.ok = false;
        }
    }
    if (!m.ok) { err = "Deserialize: " + path; return false; }
    return finish_shape(out, path, err);
}

// json / yaml / ron: {v, dim: 3 integers, data: numbers} or {pot_sub: number}, any layout.
// The text is scanned for the keys; numbers are whatever strtod accepts plus YAML's .inf/.nan.
static bool read_text_field(const std::string &path, FieldFile &out, std::string &err)
{
    std::vector<unsigned char> raw;
    if (!slurp(path, raw)) { err = "FileNotFound: " + path; return false; }
    const std::string s(raw.begin(), raw.end());
    auto key_pos = [&](const char *key) -> size_t { // `key` followed by optional quote / spaces and ':'
        const size_t kl = strlen(key);
        for (size_t p = s.find(key); p != std::string::npos; p = s.find(key, p + 1)) {
            if (p > 0 && (isalnum((unsigned char)s[p - 1]) || s[p - 1] == '_')) continue;
            size_t q = p + kl;
            while (q < s.size() && (s[q] == '"' || s[q] == ' ')) ++q;
            if (q < s.size() && s[q] == ':') return q + 1;
        }
        return std::string::npos;
    };
    auto next_number = [&](size_t &p, size_t stop, double &v) -> bool {
        while (p < stop) {
            const char c = s[p];
            if (isdigit((unsigned char)c) || ((c == '-' || c == '+' || c == '.') && p + 1 < stop &&
                                              (isdigit((unsigned char)s[p + 1]) || s[p + 1] == '.' || s[p + 1] == 'i' || s[p + 1] == 'n' || s[p + 1] == 'N'))) {
                if (s.compare(p, 4, ".inf") == 0 || s.compare(p, 5, "+.inf") == 0) { v = INFINITY; p += 4; return true; }
                if (s.compare(p, 5, "-.inf") == 0) { v = -INFINITY; p += 5; return true; }
                if (s.compare(p, 4, ".nan") == 0) { v = NAN; p += 4; return true; }
                char *endp = nullptr;
                v = strtod(s.c_str() + p, &endp);
                if (endp != s.c_str() + p) { p = (size_t)(endp - s.c_str()); return true; }
            }
            if (c == 'N' && s.compare(p, 3, "NaN") == 0) { v = NAN; p += 3; return true; }
            if (c == 'i' && s.compare(p, 3, "inf") == 0) { v = INFINITY; p += 3; return true; }
            ++p;
        }
        return false;
    };
    size_t ps = key_pos("pot_sub");
    if (ps != std::string::npos) {
        out.scalar = true;
        if (!next_number(ps, s.size(), out.value)) { err = "Deserialize: " + path; return false; }
        return true;
    }
    size_t pd = key_pos("dim"), pa = key_pos("data");
    if (pd == std::string::npos || pa == std::string::npos) { err = "Deserialize: " + path + ": no dim / data"; return false; }
    double d[3];
    const size_t dim_stop = pa > pd ? pa : s.size();
    for (int i = 0; i < 3; ++i)
        if (!next_number(pd, dim_stop, d[i])) { err = "Deserialize: " + path + ": dim"; return false; }
    out.nx = (uint32_t)d[0]; out.ny = (uint32_t)d[1]; out.nz = (uint32_t)d[2];
    const size_t data_stop = pd > pa ? s.rfind("dim") : s.size();
    out.data.reserve((size_t)out.nx * out.ny * out.nz);
    double v;
    while (next_number(pa, data_stop, v)) out.data.push_back(v);
    return finish_shape(out, path, err);
}

static bool read_field(const std::string &path, int file_type, FieldFile &out, std::string &err)
{
    out = FieldFile();
    switch (file_type) {
    case WF_MPK: return read_mpk_field(path, out, err);
    case WF_CSV: return read_csv_field(path, out, err);
    default: return read_text_field(path, out, err);
    }
}

static int type_of_path(const std::string &path)
{
    for (int t = 0; t < 5; ++t) {
        const std::string e = kFileExt[t];
        if (path.size() > e.size() && path.compare(path.size() - e.size(), e.size(), e) == 0) return t;
    }
    return -1;
}

// input.rs:75-110, 264-300, 542-575: <dir>/<stem>.{mpk,csv,json,yaml,ron}; with several present the
// configured output type arbitrates, otherwise the first in that order.  Returns the type or -1.
static int find_input(const std::string &dir, const std::string &stem, int configured, std::string &path, bool *several = nullptr)
{
    int count = 0, first = -1;
    for (int t = 0; t < 5; ++t)
        if (file_exists(dir + "/" + stem + kFileExt[t])) {
            if (first < 0) first = t;
            ++count;
        }
    if (several) *several = count > 1;
    if (count == 0) return -1;
    int pick = first;
    if (count > 1 && file_exists(dir + "/" + stem + kFileExt[configured])) pick = configured;
    path = dir + "/" + stem + kFileExt[pick];
    return pick;
}
