// wafer-hip -- host driver above the C ABI (include/wafer_hip.h).
//
// Mirrors the reference's run path for a `wafer.yaml` (main.rs:94-240 ->
// grid::run, grid.rs:31-47 -> solve, grid.rs:50-246) with the hot path on the
// GPU: same configuration keys (config.rs:292-333), same validation
// (config.rs:362-370), same per-state table (output.rs:421-521), summary
// (output.rs:559-603) and observables_N / wavefunction_N / potential outputs
// (output.rs:32-45, 85-165, 379-400, 533-677).
//
// Input and output files in all five of the reference's formats (wafer_files.h).
//
// potential: FromScript runs ./<script> (-s, default gen_potential.py) with the reference's
// stdin / stdout protocol (input.rs:186-246) before the GPU is touched.
//
// Out of scope: slog file logging, the progress bar.
//
//   wafer-hip [-c wafer.yaml] [-s SCRIPT] [--check-config] [--progress] [--output-dir DIR] [--input-dir DIR]
//   wafer-hip --convert IN OUT      (array / potential_sub file from one format to another, by extension)
//   wafer-hip --convert IN OUT.npy [--pad E]   (numpy's format, optionally framed: wafer_amd.run's input)
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <csignal>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <thread>
#include <vector>

#include "../../include/wafer_hip.h"
#include "wafer_files.h"

// ---------------------------------------------------------------------------
// A reader for the YAML subset wafer.yaml uses: nested maps by indentation,
// scalars, '#' comments.  Keys are flattened to "grid.size.x".
// ---------------------------------------------------------------------------
static std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

static bool parse_yaml(const std::string &path, std::map<std::string, std::string> &out, std::string &err)
{
    std::ifstream f(path);
    if (!f) {
        err = "ConfigLoad: cannot open " + path; // ErrorKind::ConfigLoad
        return false;
    }
    std::vector<std::pair<int, std::string>> stack; // (indent, key)
    std::string line;
    int lineno = 0;
    while (std::getline(f, line)) {
        ++lineno;
        // strip comments: '#' at line start or preceded by whitespace, outside quotes (YAML 1.2 6.6)
        char quote = 0;
        for (size_t i = 0; i < line.size(); ++i) {
            const char ch = line[i];
            if (quote) {
                if (ch == quote) quote = 0;
            } else if (ch == '"' || ch == '\'') {
                quote = ch;
            } else if (ch == '#' && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) {
                line.erase(i);
                break;
            }
        }
        if (trim(line).empty()) continue;
        const int indent = (int)line.find_first_not_of(' ');
        const std::string body = trim(line);
        const size_t colon = body.find(':');
        if (colon == std::string::npos) {
            err = "Deserialize: line " + std::to_string(lineno) + ": expected 'key: value'";
            return false;
        }
        const std::string key = trim(body.substr(0, colon));
        std::string val = trim(body.substr(colon + 1));
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        std::string full;
        for (auto &p : stack) full += p.second + ".";
        full += key;
        if (val.empty()) {
            stack.push_back({indent, key});
        } else {
            if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'') && val.back() == val.front())
                val = val.substr(1, val.size() - 2);
            out[full] = val;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------
// Config (config.rs:292-333)
// ---------------------------------------------------------------------------
static const char *kPotentials[] = {"NoPotential", "Cube", "QuadWell", "Periodic", "Coulomb", "ComplexCoulomb",
                                    "ElipticalCoulomb", "SimpleCornell", "FullCornell", "Harmonic",
                                    "ComplexHarmonic", "Dodecahedron", "FromFile", "FromScript"};
static const char *kSymmetry[] = {"NotConstrained", "AboutZ", "AntisymAboutZ", "AboutY", "AntisymAboutY"}; // config.rs:184-197
static const char *kInitialConditions[] = {"FromFile", "Gaussian", "Coulomb", "Constant", "Boolean"};

struct Config {
    std::string project_name = "wafer";
    uint32_t nx = 0, ny = 0, nz = 0;
    double dn = 0, dt = 0, tolerance = 0, mass = 1, sig = 1;
    int central_difference = 1;
    bool has_max_steps = false;
    uint64_t max_steps = 0;
    uint32_t wavenum = 0, wavemax = 0;
    int potential = 0, init_condition = 4, file_type = 1;
    std::string init_symmetry = "NotConstrained";
    uint64_t screen_update = 1000;
    bool has_snap_update = false;
    uint64_t snap_update = 0;
    bool save_wavefns = false, save_potential = false;
    std::string dtype = "f64"; // engine extension: optional `gpu.dtype`
    int device = 0;            // engine extension: optional `gpu.device`
};

static int index_of(const char *const *names, int n, const std::string &v)
{
    for (int i = 0; i < n; ++i)
        if (v == names[i]) return i;
    return -1;
}

static bool load_config(const std::string &path, Config &c, std::string &err)
{
    std::map<std::string, std::string> kv;
    if (!parse_yaml(path, kv, err)) return false;
    auto need = [&](const char *k, std::string &dst) {
        auto it = kv.find(k);
        if (it == kv.end()) {
            err = std::string("Deserialize: missing field `") + k + "`";
            return false;
        }
        dst = it->second;
        return true;
    };
    auto num = [&](const char *k, double &dst) {
        std::string s;
        if (!need(k, s)) return false;
        char *end = nullptr;
        dst = strtod(s.c_str(), &end);
        if (end == s.c_str() || *end) {
            err = std::string("Deserialize: field `") + k + "`: not a number: " + s;
            return false;
        }
        return true;
    };
    auto boolean = [&](const char *k, bool &dst) {
        std::string s;
        if (!need(k, s)) return false;
        if (s == "true") dst = true;
        else if (s == "false") dst = false;
        else {
            err = std::string("Deserialize: field `") + k + "`: expected true/false";
            return false;
        }
        return true;
    };
    std::string s;
    double d;
    if (!need("project_name", c.project_name)) return false;
    if (!num("grid.size.x", d)) return false; c.nx = (uint32_t)d;
    if (!num("grid.size.y", d)) return false; c.ny = (uint32_t)d;
    if (!num("grid.size.z", d)) return false; c.nz = (uint32_t)d;
    if (!num("grid.dn", c.dn) || !num("grid.dt", c.dt) || !num("tolerance", c.tolerance)) return false;
    if (!need("central_difference", s)) return false;
    c.central_difference = s == "ThreePoint" ? 1 : s == "FivePoint" ? 2 : s == "SevenPoint" ? 3 : 0;
    if (!c.central_difference) { err = "Deserialize: unknown central_difference `" + s + "`"; return false; }
    if (kv.count("max_steps")) { if (!num("max_steps", d)) return false; c.has_max_steps = true; c.max_steps = (uint64_t)d; }
    if (!num("wavenum", d)) return false; c.wavenum = (uint32_t)d;
    if (!num("wavemax", d)) return false; c.wavemax = (uint32_t)d;
    if (!need("potential", s)) return false;
    if ((c.potential = index_of(kPotentials, 14, s)) < 0) { err = "Deserialize: unknown potential `" + s + "`"; return false; }
    if (!num("mass", c.mass)) return false;
    if (!need("init_condition", s)) return false;
    if ((c.init_condition = index_of(kInitialConditions, 5, s)) < 0) { err = "Deserialize: unknown init_condition `" + s + "`"; return false; }
    if (!num("sig", c.sig)) return false;
    if (!need("init_symmetry", c.init_symmetry)) return false;
    if (index_of(kSymmetry, 5, c.init_symmetry) < 0) { err = "init_symmetry: unknown variant `" + c.init_symmetry + "`"; return false; }
    if (!num("output.screen_update", d)) return false; c.screen_update = (uint64_t)d;
    if (kv.count("output.snap_update")) { if (!num("output.snap_update", d)) return false; c.has_snap_update = true; c.snap_update = (uint64_t)d; }
    if (!need("output.file_type", s)) return false;
    if ((c.file_type = index_of(kFileTypes, 5, s)) < 0) { err = "Deserialize: unknown file_type `" + s + "`"; return false; }
    if (!boolean("output.save_wavefns", c.save_wavefns) || !boolean("output.save_potential", c.save_potential)) return false;
    if (kv.count("gpu.dtype")) c.dtype = kv["gpu.dtype"];
    if (kv.count("gpu.device")) c.device = atoi(kv["gpu.device"].c_str());
    // Config::parse, config.rs:362-370
    if (c.dt > c.dn * c.dn / 3.) { err = "ConfigParse: LargeDt: dt must be <= dn^2/3"; return false; }
    if (c.wavenum > c.wavemax) { err = "ConfigParse: LargeWavenum: wavenum must be <= wavemax"; return false; }
    if (c.dtype != "f64" && c.dtype != "f32" && c.dtype != "f32fast") { err = "gpu.dtype must be f64, f32 or f32fast"; return false; }
    return true;
}

// ---------------------------------------------------------------------------
// number formatting the way Rust's std::fmt does it
// ---------------------------------------------------------------------------
// `{:.Ne}` on f64: d.ddd…e<exp> with a bare exponent ("1.5000000000e0", "3.2e-5")
static std::string rust_lower_exp(double v, int prec)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    snprintf(buf, sizeof buf, "%.*e", prec, v);
    std::string s(buf);
    const size_t e = s.find('e');
    const int ex = atoi(s.c_str() + e + 1);
    return s.substr(0, e + 1) + std::to_string(ex);
}
static std::string pad_left(const std::string &s, size_t w) { return s.size() >= w ? s : std::string(w - s.size(), ' ') + s; }
static std::string fixed(double v, int prec)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.*f", prec, v);
    return buf;
}
static size_t utf8_len(const std::string &s)
{
    size_t n = 0;
    for (unsigned char ch : s) n += (ch & 0xC0) != 0x80;
    return n;
}
// `{:F^w$}`: centre `s` in w columns with fill (extra fill goes to the right)
static std::string centre(const std::string &s, size_t w, const std::string &fill)
{
    const size_t len = utf8_len(s);
    if (len >= w) return s;
    const size_t total = w - len, left = total / 2, right = total - left;
    std::string out;
    for (size_t i = 0; i < left; ++i) out += fill;
    out += s;
    for (size_t i = 0; i < right; ++i) out += fill;
    return out;
}
static std::string ordinal(unsigned n)
{
    const char *suf = "th";
    if (n % 100 < 11 || n % 100 > 13) {
        if (n % 10 == 1) suf = "st";
        else if (n % 10 == 2) suf = "nd";
        else if (n % 10 == 3) suf = "rd";
    }
    return std::to_string(n) + suf;
}

static const size_t kTermWidth = 100; // output.rs:733-745 without a terminal

// output.rs:421-494
static void print_observable_header(unsigned wnum)
{
    const size_t width = kTermWidth, spacer = (width - 69) / 2;
    const size_t rspace = (2 * spacer + 69 < width) ? spacer + 1 : spacer;
    const std::string title = wnum == 0 ? " Ground state caclulation " : " " + ordinal(wnum) + " excited state caclulation ";
    printf("\n%s╤%s╤%s╤%s╤%s\n", centre("", spacer, "═").c_str(), centre("", 12, "═").c_str(),
           centre(title, 37, "═").c_str(), centre("", 16, "═").c_str(), centre("", rspace, "═").c_str());
    printf("%s│%s│%s│%s│%s│\n", centre("", spacer, " ").c_str(), centre("Time (τ)", 12, " ").c_str(),
           centre("Energy", 20, " ").c_str(), centre("rᵣₘₛ", 16, " ").c_str(), centre("Difference", 16, " ").c_str());
    printf("%s┼%s┼%s┼%s┼%s┼%s\n", centre("", spacer, "─").c_str(), centre("", 12, "─").c_str(),
           centre("", 20, "─").c_str(), centre("", 16, "─").c_str(), centre("", 16, "─").c_str(),
           centre("", rspace, "─").c_str());
}

// output.rs:497-521
static std::string format_measurements(double tau, double diff, const wafer_observables_t &o)
{
    const size_t spacer = (kTermWidth - 69) / 2;
    std::string s = std::string(spacer, ' ') + "│" + pad_left(fixed(tau, 3), 11) + " │" +
                    pad_left(rust_lower_exp(o.energy / o.norm2, 10), 19) + " │" +
                    pad_left(fixed(std::sqrt(o.r2 / o.norm2), 5), 15) + " │";
    s += (tau > 0.0) ? pad_left(rust_lower_exp(diff, 5), 15) + " │" : pad_left("--   ", 15) + " │";
    return s;
}

// output.rs:559-603
static void print_summary(const wafer_observables_output &o)
{
    const size_t width = kTermWidth, spacer = (width - 69) / 2;
    const size_t rspace = (2 * spacer + 69 < width) ? spacer + 1 : spacer;
    printf("%s╧%s╧%s╧%s╧%s╧%s\n", centre("", spacer, "═").c_str(), centre("", 12, "═").c_str(),
           centre("", 20, "═").c_str(), centre("", 16, "═").c_str(), centre("", 16, "═").c_str(),
           centre("", rspace, "═").c_str());
    const std::string who = o.state == 0 ? "Ground state" : ordinal(o.state) + " excited state";
    printf("══▶ %s energy = %s\n", who.c_str(), rust_display(o.energy).c_str());
    printf("══▶ %s binding energy = %s\n", who.c_str(), rust_display(o.binding_energy).c_str());
    printf("══▶ rᵣₘₛ = %s\n", rust_display(o.r).c_str());
    printf("══▶ L/rᵣₘₛ = %s\n\n", rust_display(o.l_r).c_str());
}

// ---------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------
// embeds an unpadded array of the configured size into the zero frame (input.rs:644-650)
static std::vector<double> embed(const FieldFile &a, uint32_t e)
{
    const size_t py = a.ny + 2 * e, pz = a.nz + 2 * e;
    std::vector<double> p((size_t)(a.nx + 2 * e) * py * pz, 0.0);
    for (uint32_t i = 0; i < a.nx; ++i)
        for (uint32_t j = 0; j < a.ny; ++j)
            memcpy(&p[((size_t)(i + e) * py + (j + e)) * pz + e], &a.data[((size_t)i * a.ny + j) * a.nz], sizeof(double) * a.nz);
    return p;
}

// <input_dir>/<stem>.* in whichever format is present (wafer_files.h find_input)
static bool load_input(const std::string &dir, const std::string &stem, int configured, FieldFile &out, std::string &err)
{
    std::string path;
    bool several = false;
    const int t = find_input(dir, stem, configured, path, &several);
    if (t < 0) { err = "FileNotFound: " + dir + "/" + stem + ".*"; return false; }
    if (several)
        fprintf(stderr, "Warning: multiple %s files found in input directory. Choosing '%s'.\n", stem.c_str(), path.c_str());
    return read_field(path, t, out, err);
}

// input::script_potential (input.rs:186-246): spawn ./<script>, write {"grid":{"dn":..,"x":..,"y":..,"z":..}}
// (serde_json's key order) to its stdin, close it, parse one f64 per stdout line -- nx*ny*nz values of
// the WORK area, x slowest.  Runs before the GPU is touched: a process that has initialised HIP must
// not fork + exec on the GPU boxes.
static bool run_potential_script(const std::string &file, const Config &cfg, FieldFile &out, std::string &err)
{
    int to_child[2], from_child[2];
    if (pipe(to_child) != 0 || pipe(from_child) != 0) { err = "SpawnPython: pipe failed"; return false; }
    const pid_t pid = fork();
    if (pid < 0) { err = "SpawnPython: Unable to spawn a python script process"; return false; }
    if (pid == 0) {
        dup2(to_child[0], 0);
        dup2(from_child[1], 1);
        close(to_child[0]); close(to_child[1]); close(from_child[0]); close(from_child[1]);
        execl(file.c_str(), file.c_str(), (char *)nullptr);
        _exit(127);
    }
    close(to_child[0]);
    close(from_child[1]);
    const std::string input = "{\"grid\":{\"dn\":" + num_text(cfg.dn) + ",\"x\":" + std::to_string(cfg.nx) + ",\"y\":" +
                              std::to_string(cfg.ny) + ",\"z\":" + std::to_string(cfg.nz) + "}}";
    signal(SIGPIPE, SIG_IGN);
    const bool wrote = write(to_child[1], input.data(), input.size()) == (ssize_t)input.size();
    close(to_child[1]); // the script starts processing once its stdin is closed
    std::string text;
    char buf[1 << 16];
    for (ssize_t n; (n = read(from_child[0], buf, sizeof buf)) > 0;) text.append(buf, (size_t)n);
    close(from_child[0]);
    int status = 0;
    waitpid(pid, &status, 0);
    if (WIFEXITED(status) && WEXITSTATUS(status) == 127 && text.empty()) { err = "SpawnPython: Unable to spawn a python script process (" + file + ")"; return false; }
    if (!wrote) { err = "StdIn: Unable to write to stdin in of the python script process"; return false; }
    out = FieldFile();
    out.nx = cfg.nx; out.ny = cfg.ny; out.nz = cfg.nz;
    size_t pos = 0;
    while (pos < text.size()) { // str::lines + parse::<f64>: the whole line must be a float
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = eol + 1;
        char *endp = nullptr;
        const double v = strtod(line.c_str(), &endp); // Rust's parse::<f64> takes no surrounding blanks
        if (line.empty() || isspace((unsigned char)line[0]) || endp != line.c_str() + line.size()) {
            err = "ParseFloat: Cannot parse float '" + line + "'";
            return false;
        }
        out.data.push_back(v);
    }
    const size_t want = (size_t)cfg.nx * cfg.ny * cfg.nz;
    if (out.data.size() != want) {
        err = "ArrayShape: Cannot reshape " + std::to_string(out.data.size()) + " values into [" + std::to_string(cfg.nx) + ", " +
              std::to_string(cfg.ny) + ", " + std::to_string(cfg.nz) + "]";
        return false;
    }
    return true;
}

// what=0: phi, 1: potential.  Same size -> copied; otherwise trilinearly resampled on the
// device with the reference's basis (input.rs:651-655, 667-716).
static int upload_field(wafer_ctx *ctx, const Config &cfg, const FieldFile &a, int what)
{
    const uint32_t e = (uint32_t)cfg.central_difference;
    if (a.nx == cfg.nx && a.ny == cfg.ny && a.nz == cfg.nz) {
        const std::vector<double> p = embed(a, e);
        return what == 0 ? wafer_upload_phi(ctx, p.data())
                         : wafer_set_potential_host(ctx, p.data(), WAFER_POTSUB_NONE, 0.0, nullptr);
    }
    fprintf(stderr, "Interpolating from [%u, %u, %u] to requested size of [%u, %u, %u] (size includes central difference padding).\n",
            a.nx + 2 * e, a.ny + 2 * e, a.nz + 2 * e, cfg.nx + 2 * e, cfg.ny + 2 * e, cfg.nz + 2 * e);
    return what == 0 ? wafer_upload_phi_resampled(ctx, a.data.data(), a.nx, a.ny, a.nz, nullptr)
                     : wafer_set_potential_resampled(ctx, a.data.data(), a.nx, a.ny, a.nz, nullptr);
}

// output.rs:605-677
static bool write_observables(const std::string &dir, int file_type, const wafer_observables_output &o, std::string &err)
{
    const std::string path = dir + "/observables_" + std::to_string(o.state) + kFileExt[file_type];
    if (file_type == WF_MPK) { // ObservablesOutput as a 5-array (output.rs:606-617)
        MpkOut m;
        m.array_len(5);
        m.uint(o.state);
        m.f64(o.energy); m.f64(o.binding_energy); m.f64(o.r); m.f64(o.l_r);
        if (!m.save(path)) { err = "CreateFile: " + path; return false; }
        return true;
    }
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { err = "CreateFile: " + path; return false; }
    const std::string e = num_text(o.energy), b = num_text(o.binding_energy), r = num_text(o.r), l = num_text(o.l_r);
    switch (file_type) {
    case 1: fprintf(f, "state,energy,binding_energy,r,l_r\n%u,%s,%s,%s,%s\n", o.state, e.c_str(), b.c_str(), r.c_str(), l.c_str()); break;
    case 2: fprintf(f, "{\n  \"state\": %u,\n  \"energy\": %s,\n  \"binding_energy\": %s,\n  \"r\": %s,\n  \"l_r\": %s\n}", o.state, e.c_str(), b.c_str(), r.c_str(), l.c_str()); break;
    case 3: fprintf(f, "---\nstate: %u\nenergy: %s\nbinding_energy: %s\nr: %s\nl_r: %s\n", o.state, e.c_str(), b.c_str(), r.c_str(), l.c_str()); break;
    default: fprintf(f, "(\n    state: %u,\n    energy: %s,\n    binding_energy: %s,\n    r: %s,\n    l_r: %s,\n)", o.state, e.c_str(), b.c_str(), r.c_str(), l.c_str()); break;
    }
    fclose(f);
    return true;
}

// sanitize_string, output.rs:722-745: letters, digits, '-', '_' and a non-leading '.' stay,
// a space becomes '_', anything else ",<code point>,"
static std::string sanitize(const std::string &s)
{
    std::string out;
    for (size_t i = 0; i < s.size(); ++i) {
        const unsigned char ch = (unsigned char)s[i];
        const bool ok = isalnum(ch) || ch == '-' || ch == '_' || (ch == '.' && i != 0);
        if (ok) out += (char)ch;
        else if (ch == ' ') out += '_';
        else out += "," + std::to_string((int)ch) + ",";
    }
    return out;
}

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != WAFER_OK) {                                                      \
            fprintf(stderr, "Error: %s\n  caused by: %s\n", #call, wafer_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char **argv)
{
    std::string config_file = "wafer.yaml", output_root = "./output", input_dir = "./input", script_file = "gen_potential.py";
    bool check_only = false, progress = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "-c" || a == "--config") && i + 1 < argc) config_file = argv[++i];
        else if (a == "--output-dir" && i + 1 < argc) output_root = argv[++i];
        else if (a == "--input-dir" && i + 1 < argc) input_dir = argv[++i];
        else if (a == "--check-config") check_only = true;
        else if (a == "--sanitize" && i + 1 < argc) { // exposes sanitize_string for its reference test vector
            printf("%s\n", sanitize(argv[++i]).c_str());
            return 0;
        }
        else if (a == "--convert" && i + 2 < argc) { // any of the five formats to any other (wafer_files.h)
            const std::string in = argv[i + 1], out = argv[i + 2];
            const int ti = type_of_path(in), to = type_of_path(out);
            const bool to_npy = out.size() > 4 && out.compare(out.size() - 4, 4, ".npy") == 0;
            FieldFile f;
            std::string e;
            if (ti < 0 || (to < 0 && !to_npy)) { fprintf(stderr, "Error: unknown file extension\n"); return 2; }
            if (!read_field(in, ti, f, e)) { fprintf(stderr, "Error: %s\n", e.c_str()); return 1; }
            if (to_npy) { // numpy's format, optionally with the zero frame of `--pad E` cells: what
                          // wafer_amd.run memory-maps so that each rank touches only its own planes
                uint32_t pad = 0;
                if (i + 4 < argc && std::string(argv[i + 3]) == "--pad") pad = (uint32_t)atoi(argv[i + 4]);
                if (!write_npy(out, f, pad, e)) { fprintf(stderr, "Error: %s\n", e.c_str()); return 1; } // a single value: shape ()
                return 0;
            }
            const bool ok = f.scalar ? write_scalar_sub(out, to, f.value, e)
                                     : write_array(out, to, f.data.data(), f.nx, f.ny, f.nz, 0, e);
            if (!ok) { fprintf(stderr, "Error: %s\n", e.c_str()); return 1; }
            return 0;
        }
        else if ((a == "-s" || a == "--script") && i + 1 < argc) script_file = argv[++i];
        else if (a == "--progress") progress = true;
        else if (a == "-h" || a == "--help") {
            printf("wafer-hip [-c wafer.yaml] [-s gen_potential.py] [--check-config] [--progress] [--output-dir DIR] [--input-dir DIR]\n"
                   "wafer-hip --convert IN OUT   (.mpk .csv .json .yaml .ron; OUT may be .npy, then [--pad E] adds the zero frame)\n");
            return 0;
        } else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    Config cfg;
    std::string err;
    if (!load_config(config_file, cfg, err)) {
        fprintf(stderr, "Error: %s\n", err.c_str());
        return 1;
    }
    if (check_only) { // machine-readable echo of what was parsed (used by the CPU tests)
        printf("{\"project_name\": \"%s\", \"nx\": %u, \"ny\": %u, \"nz\": %u, \"dn\": %s, \"dt\": %s, \"tolerance\": %s, "
               "\"central_difference\": %d, \"max_steps\": %s, \"wavenum\": %u, \"wavemax\": %u, \"potential\": \"%s\", "
               "\"mass\": %s, \"init_condition\": \"%s\", \"sig\": %s, \"init_symmetry\": \"%s\", \"screen_update\": %llu, "
               "\"snap_update\": %s, \"file_type\": \"%s\", \"save_wavefns\": %s, \"save_potential\": %s, \"dtype\": \"%s\"}\n",
               cfg.project_name.c_str(), cfg.nx, cfg.ny, cfg.nz, num_text(cfg.dn).c_str(), num_text(cfg.dt).c_str(),
               num_text(cfg.tolerance).c_str(), cfg.central_difference,
               cfg.has_max_steps ? std::to_string(cfg.max_steps).c_str() : "null", cfg.wavenum, cfg.wavemax,
               kPotentials[cfg.potential], num_text(cfg.mass).c_str(), kInitialConditions[cfg.init_condition],
               num_text(cfg.sig).c_str(), cfg.init_symmetry.c_str(), (unsigned long long)cfg.screen_update,
               cfg.has_snap_update ? std::to_string(cfg.snap_update).c_str() : "null", kFileTypes[cfg.file_type],
               cfg.save_wavefns ? "true" : "false", cfg.save_potential ? "true" : "false", cfg.dtype.c_str());
        return 0;
    }
    const int symmetry = index_of(kSymmetry, 5, cfg.init_symmetry);
    if (symmetry != WAFER_SYM_NOT_CONSTRAINED && cfg.central_difference != WAFER_CD_SEVENPOINT) {
        // the reference would panic with an out-of-bounds index here (config.rs:702-725 walks n + 6 cells)
        fprintf(stderr, "Error: init_symmetry %s needs central_difference: SevenPoint (config.rs:702-725 indexes the 3-cell frame)\n",
                cfg.init_symmetry.c_str());
        return 1;
    }
    FieldFile scripted; // potential.rs:87-94; config.rs:344-347: the script lives at ./<name>
    if (cfg.potential == WAFER_POT_FROMSCRIPT) {
        const std::string file = "./" + script_file;
        fprintf(stderr, "Generating potential from script file: %s\n", file.c_str());
        if (!run_potential_script(file, cfg, scripted, err)) {
            fprintf(stderr, "Error: LoadPotential: %s\n", err.c_str());
            return 1;
        }
    }

    // output directory ./output/<project>_<timestamp> (output.rs:680-699)
    char stamp[64];
    time_t now = time(nullptr);
    strftime(stamp, sizeof stamp, "%Y-%m-%d_%H:%M:%S", localtime(&now));
    mkdir(output_root.c_str(), 0777);
    const std::string out_dir = output_root + "/" + sanitize(cfg.project_name) + "_" + stamp;
    if (mkdir(out_dir.c_str(), 0777) != 0) { fprintf(stderr, "Error: CreateOutputDir %s\n", out_dir.c_str()); return 1; }
    { // copy the configuration next to the results (output.rs:702-706)
        std::ifstream src(config_file, std::ios::binary);
        std::ofstream dst(out_dir + "/" + config_file.substr(config_file.find_last_of('/') + 1), std::ios::binary);
        dst << src.rdbuf();
    }

    const uint32_t e = (uint32_t)cfg.central_difference;
    wafer_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.nx = cfg.nx; p.ny = cfg.ny; p.nz = cfg.nz;
    p.central_difference = cfg.central_difference;
    p.dtype = cfg.dtype == "f32" ? WAFER_F32 : cfg.dtype == "f32fast" ? WAFER_F32_FAST : WAFER_F64;
    p.dn = cfg.dn; p.dt = cfg.dt; p.mass = cfg.mass; p.sig = cfg.sig;
    p.max_states = cfg.wavemax + 1;
    p.device = cfg.device;
    wafer_ctx *ctx = nullptr;
    CHECK(wafer_ctx_create(&p, &ctx));

    const size_t padded_len = (size_t)(cfg.nx + 2 * e) * (cfg.ny + 2 * e) * (cfg.nz + 2 * e);
    std::vector<double> host;
    const std::string ext = kFileExt[cfg.file_type];

    // potential::load_arrays (potential.rs:75-175)
    if (cfg.potential == WAFER_POT_FROMFILE) {
        FieldFile pot;
        if (!load_input(input_dir, "potential", cfg.file_type, pot, err) || pot.scalar) {
            fprintf(stderr, "Error: LoadPotential: %s\n", pot.scalar ? "potential file holds a single value" : err.c_str());
            return 1;
        }
        CHECK(upload_field(ctx, cfg, pot, 1));
    } else if (cfg.potential == WAFER_POT_FROMSCRIPT) {
        CHECK(upload_field(ctx, cfg, scripted, 1)); // "generated is the right size by definition: copy down"
        scripted = FieldFile();
    } else {
        CHECK(wafer_set_potential_builtin(ctx, cfg.potential));
    }
    { // a potential_sub file in ./input overrides the computed one (potential.rs:113-131)
        FieldFile sub;
        std::string sub_err;
        if (load_input(input_dir, "potential_sub", cfg.file_type, sub, sub_err)) {
            const bool variable = cfg.potential == WAFER_POT_FULLCORNELL; // PotentialType::variable_pot_sub
            if (sub.scalar && variable) {
                fprintf(stderr, "Error: WrongPotentialSubDims: potential_sub input file contains a singular value, but potential type is FullCornell.\n");
                return 1;
            }
            if (!sub.scalar && !variable) {
                fprintf(stderr, "Error: WrongPotentialSubDims: potential_sub input file contains an array, but potential type is not FullCornell.\n");
                return 1;
            }
            if (sub.scalar) {
                CHECK(wafer_set_potsub(ctx, WAFER_POTSUB_SCALAR, sub.value, nullptr));
            } else {
                if (sub.nx != cfg.nx || sub.ny != cfg.ny || sub.nz != cfg.nz) { // input::fill_sub_data (input.rs:453-478)
                    fprintf(stderr, "Interpolating potential_sub from [%u, %u, %u] to requested size of [%u, %u, %u].\n", sub.nx,
                            sub.ny, sub.nz, cfg.nx, cfg.ny, cfg.nz);
                    CHECK(wafer_set_potsub_resampled(ctx, sub.data.data(), sub.nx, sub.ny, sub.nz));
                } else {
                    CHECK(wafer_set_potsub(ctx, WAFER_POTSUB_ARRAY, 0.0, sub.data.data()));
                }
            }
            fprintf(stderr, "Potential_sub loaded from disk\n");
        }
    }
    if (cfg.save_potential) {
        host.resize(padded_len);
        CHECK(wafer_download_array(ctx, WAFER_ARRAY_V, host.data()));
        if (!write_array(out_dir + "/potential" + ext, cfg.file_type, host.data(), cfg.nx, cfg.ny, cfg.nz, e, err))
            fprintf(stderr, "Warning: could not write potential to disk: %s\n", err.c_str());
        // output::potential_sub (output.rs:103-141): the array for FullCornell, a positive scalar, else nothing
        int kind = 0;
        double scalar = 0.0;
        CHECK(wafer_get_potsub(ctx, &kind, &scalar));
        bool ok = true;
        if (kind == WAFER_POTSUB_ARRAY) {
            std::vector<double> sub((size_t)cfg.nx * cfg.ny * cfg.nz);
            CHECK(wafer_download_array(ctx, WAFER_ARRAY_POTSUB, sub.data()));
            ok = write_array(out_dir + "/potential_sub" + ext, cfg.file_type, sub.data(), cfg.nx, cfg.ny, cfg.nz, 0, err);
        } else if (kind == WAFER_POTSUB_SCALAR && scalar > 0.0) {
            ok = write_scalar_sub(out_dir + "/potential_sub" + ext, cfg.file_type, scalar, err);
        }
        if (!ok) fprintf(stderr, "Warning: could not write potential_sub to disk: %s\n", err.c_str());
    }
    // grid.rs:35-39: converged lower states must come from disk when wavenum > 0
    for (uint32_t w = 0; w < cfg.wavenum; ++w) {
        FieldFile st;
        if (!load_input(input_dir, "wavefunction_" + std::to_string(w), cfg.file_type, st, err) || st.scalar) {
            fprintf(stderr, "Error: LoadWavefunction(%u): %s\n", w, err.c_str());
            return 1;
        }
        CHECK(upload_field(ctx, cfg, st, 0));
        CHECK(wafer_push_state(ctx));
    }

    struct Joined { // joins on every way out of main
        std::thread t;
        ~Joined() { if (t.joinable()) t.join(); }
    } writer; // at most one snapshot is being written at a time
    std::thread &snapshot_writer = writer.t;
    const clock_t t_start = clock();
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    int exit_code = 0;
    for (uint32_t wnum = cfg.wavenum; wnum <= cfg.wavemax; ++wnum) { // grid.rs:43-45
        // starting wavefunction, grid.rs:60-100
        FieldFile start;   // wavefunction_N, else wavefunction_N_partial (input.rs:513-523)
        const bool from_disk = (load_input(input_dir, "wavefunction_" + std::to_string(wnum), cfg.file_type, start, err) ||
                                load_input(input_dir, "wavefunction_" + std::to_string(wnum) + "_partial", cfg.file_type, start, err)) &&
                               !start.scalar;
        bool cloned = false;
        if (wnum > 0) {
            if (from_disk) CHECK(upload_field(ctx, cfg, start, 0));
            else {
                CHECK(wafer_clone_state_to_phi(ctx, wnum - 1));
                cloned = true;
            }
        } else if (cfg.init_condition == WAFER_IC_FROMFILE) {
            if (!from_disk) { fprintf(stderr, "Error: SetInitialConditions: LoadWavefunction(0): %s\n", err.c_str()); return 1; }
            CHECK(upload_field(ctx, cfg, start, 0));
        } else {
            CHECK(wafer_set_initial_condition(ctx, cfg.init_condition, (uint64_t)now));
        }
        if (wnum == 0) CHECK(wafer_symmetrise(ctx, symmetry)); // config::set_initial_conditions, config.rs:625
        print_observable_header(wnum);
        struct timespec ts_state;
        clock_gettime(CLOCK_MONOTONIC, &ts_state);
        // solve, grid.rs:122-246 (the loop itself: wafer_solve_state == grid.rs:126-221)
        std::vector<wafer_block_record> recs(progress ? 1u << 20 : 4);
        // run block by block so rows can be shown as they are produced
        uint64_t step = 0;
        double last_energy = 1.7976931348623157e308;
        bool converged = false;
        wafer_observables_t obs;
        for (;;) {
            CHECK(wafer_observables(ctx, &obs));
            const double norm_energy = obs.energy / obs.norm2;
            const double tau = (double)step * cfg.dt;
            CHECK(wafer_normalise(ctx, obs.norm2));
            if (wnum > 0) CHECK(wafer_orthogonalise(ctx, wnum));
            if (wnum > 0 && step == 0 && cloned) {
                // The reference starts an excited state from a CLONE of the previous one (grid.rs:95)
                // and lets Gram-Schmidt reduce it to rounding noise.  With this engine's deterministic
                // sums the overlap can round to exactly 1 and the state to exactly 0 (then 0/0 = NaN on
                // the next normalise; the reference's debug build would panic in R64).  Seed with noise
                // instead of spinning on NaNs.
                double n2 = 0.0;
                CHECK(wafer_norm2(ctx, &n2));
                if (!(n2 > 0.0) || !std::isfinite(n2)) {
                    fprintf(stderr, "Warning: the clone of state %u was annihilated exactly by Gram-Schmidt; "
                                    "starting state %u from Gaussian noise instead.\n", wnum - 1, wnum);
                    CHECK(wafer_set_initial_condition(ctx, WAFER_IC_GAUSSIAN, 0x5eedULL + wnum));
                    cloned = false;
                    last_energy = 1.7976931348623157e308;
                    continue;
                }
            }
            if (cfg.has_snap_update && step % cfg.snap_update == 0) { // grid.rs:137-158, WITHOUT its second, stale-norm2 normalise
                CHECK(wafer_symmetrise(ctx, symmetry)); // grid.rs:138
                // the copy to the host is the only part the GPU waits for; formatting and writing
                // the file (minutes for a 512^3 csv) go to a writer thread while evolve continues
                if (snapshot_writer.joinable()) snapshot_writer.join();
                std::vector<double> snap(padded_len);
                CHECK(wafer_download_phi(ctx, snap.data()));
                const std::string name = out_dir + "/wavefunction_" + std::to_string(wnum) + "_partial" + ext;
                snapshot_writer = std::thread([snap = std::move(snap), name, &cfg, e]() {
                    std::string werr;
                    if (!write_array(name + ".tmp", cfg.file_type, snap.data(), cfg.nx, cfg.ny, cfg.nz, e, werr) ||
                        rename((name + ".tmp").c_str(), name.c_str()) != 0)
                        fprintf(stderr, "Warning: could not output partial wavefunction: %s\n", werr.c_str());
                });
            }
            const double diff = std::fabs(norm_energy - last_energy);
            if (!std::isfinite(norm_energy)) { fprintf(stderr, "Error: state %u: energy is not finite at step %llu\n", wnum, (unsigned long long)step); return 1; }
            if (diff < cfg.tolerance) {
                printf("%s\n", format_measurements(tau, diff, obs).c_str());
                converged = true;
                break;
            }
            if (progress) printf("%s\n", format_measurements(tau, diff, obs).c_str());
            last_energy = norm_energy;
            if (cfg.has_max_steps && step > cfg.max_steps) break;
            CHECK(wafer_evolve(ctx, wnum, cfg.screen_update));
            step += cfg.screen_update;
        }
        if (snapshot_writer.joinable()) snapshot_writer.join();
        { // per-state accounting on stderr (the reference prints the total only)
            struct timespec ts_now;
            clock_gettime(CLOCK_MONOTONIC, &ts_now);
            const double st = (ts_now.tv_sec - ts_state.tv_sec) + 1e-9 * (ts_now.tv_nsec - ts_state.tv_nsec);
            fprintf(stderr, "state %u: %llu steps in %.3f s (%.4f ms per step, evolve + observables + normalise)\n", wnum,
                    (unsigned long long)step, st, step ? 1e3 * st / (double)step : 0.0);
        }
        wafer_observables_output fin;
        const double r_norm = std::sqrt(obs.r2 / obs.norm2);
        fin.state = wnum;
        fin.energy = obs.energy / obs.norm2;
        fin.binding_energy = (obs.energy - obs.v_infinity) / obs.norm2;
        fin.r = r_norm;
        fin.l_r = (double)cfg.nx / r_norm;
        if (converged) {
            print_summary(fin); // output::finalise_measurement, output.rs:533-558
            if (!write_observables(out_dir, cfg.file_type, fin, err)) { fprintf(stderr, "Error: SaveObservables: %s\n", err.c_str()); return 1; }
            if (cfg.has_snap_update) remove((out_dir + "/wavefunction_" + std::to_string(wnum) + "_partial" + ext).c_str());
        }
        if (cfg.save_wavefns) { // grid.rs:223-237: saved whether converged or not
            host.resize(padded_len);
            CHECK(wafer_download_phi(ctx, host.data()));
            const std::string name = out_dir + "/wavefunction_" + std::to_string(wnum) + (converged ? "" : "_partial") + ext;
            if (!write_array(name, cfg.file_type, host.data(), cfg.nx, cfg.ny, cfg.nz, e, err))
                fprintf(stderr, "Warning: could not write wavefunction to disk: %s\n", err.c_str());
        }
        if (!converged) { // grid.rs:243-245
            fprintf(stderr, "Error: MaxStep: maximum step limit reached for state %u\n", wnum);
            exit_code = 1;
            break;
        }
        CHECK(wafer_push_state(ctx)); // grid.rs:241
    }
    (void)t_start;
    struct timespec ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    const double secs = (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec);
    printf("Simulation complete. Elapsed time: %.3f seconds.\nOutput directory: %s\n", secs, out_dir.c_str());
    wafer_ctx_destroy(ctx);
    return exit_code;
}
