// psp_process, phase 1, as a C++ host program over the C ABI of libupsp_gpu.so -- no Python, no torch.
//
//   psp_process_cpp -input_file=run.inp -h5_out=out.h5 [-frames=N] [-add_out_dir=DIR] [-cutoff_x_max=X] [-ranks=N]
//
// The reference's command line (cv::CommandLineParser `-name=value`, cpp/exec/psp_process.cpp:1193-1218) and input deck
// (@general / @vars / @all / @camera / @options / @output with $var substitution, cpp/lib/upsp_inputs.cpp) for the hot
// path of phase 1 (cpp/exec/psp_process.cpp:1438-2040):
//
//   createBVH -> per camera create_projection_mat -> adjust_projection_for_weights -> identify_skipped_nodes ->
//   first-frame solution -> frame loop (read 12-bit MRAW, unpack, fix_hot_pixels, [register], [filter], project, accumulate)
//   -> MPI_Reduce / Bcast of the sums -> finals -> global_transpose -> flat files (:524-540)
//
// Inputs this driver reads: Cart3D `.tri` grids (cpp/lib/TriModel.ipp:117-257), camera calibration JSON
// (cpp/lib/CameraCal.cpp:19-54), Photron `.mraw` + `.cih` (cpp/lib/MrawReader.cpp).  PLOT3D grids, CINE files, the target
// patcher's phase 0 and phase 2 are served by the Python driver (upsp_processing_amd/psp_process.py), which this program is
// byte-compared with (tests/test_cli.py).  `-ranks=N` starts N rank processes (the reference's `mpiexec -n N`): frames shard
// with apportion() (:1519-1529), the sums go through upsp_allreduce_sums and the time series through upsp_exchange_* (RCCL).
#include <cerrno>
#include <csignal>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "upsp_gpu.h"

namespace {

struct DeckError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define CHECK(call)                                                                                             \
    do {                                                                                                        \
        const int rc_ = (call);                                                                                 \
        if (rc_ != 0) throw std::runtime_error(std::string(#call) + ": " + upsp_last_error());                  \
    } while (0)
#define HIPCHECK(call)                                                                                          \
    do {                                                                                                        \
        const hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

std::string trim(const std::string &s)
{
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

bool ends_with(const std::string &s, const std::string &t) { return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }

// ---- flags: -name=value or --name=value; a bare -name is "true" ---------------------------------------------------------
std::map<std::string, std::string> parse_flags(int argc, char **argv)
{
    std::map<std::string, std::string> f;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a.empty() || a[0] != '-') throw DeckError("unexpected argument '" + a + "'");
        a = a.substr(a.find_first_not_of('-'));
        const size_t eq = a.find('=');
        const std::string k = a.substr(0, eq), v = eq == std::string::npos ? "" : a.substr(eq + 1);
        f[k] = v.empty() ? "true" : v;
    }
    if (!f.count("input_file")) throw DeckError("missing required flag -input_file");
    return f;
}

// ---- input deck (FileInputs::Load, cpp/lib/upsp_inputs.cpp) --------------------------------------------------------------
struct Deck {
    std::map<std::string, std::string> general, all, options, output;
    std::vector<std::pair<std::string, std::string>> vars;           // in file order, like the Python driver's dict
    std::vector<std::map<std::string, std::string>> cameras;
};

Deck parse_deck(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw DeckError("cannot open input file " + path);
    Deck d;
    std::string section, raw;
    while (std::getline(in, raw)) {
        const std::string line = trim(raw.substr(0, raw.find('#')));
        if (line.empty()) continue;
        if (line[0] == '@') {
            section = trim(line.substr(1));
            std::transform(section.begin(), section.end(), section.begin(), ::tolower);
            if (section != "general" && section != "vars" && section != "all" && section != "camera" && section != "options" &&
                section != "output")
                throw DeckError("unknown section @" + section);
            if (section == "camera") d.cameras.emplace_back();
            continue;
        }
        const size_t eq = line.find('=');
        if (section.empty() || eq == std::string::npos) throw DeckError("malformed line '" + trim(raw) + "'");
        const std::string k = trim(line.substr(0, eq));
        std::string v = trim(line.substr(eq + 1));
        for (const auto &nv : d.vars) {                               // $var substitution
            const std::string key = "$" + nv.first;
            for (size_t p = v.find(key); p != std::string::npos; p = v.find(key, p + nv.second.size())) v.replace(p, key.size(), nv.second);
        }
        if (section == "vars") {
            bool found = false;
            for (auto &nv : d.vars)
                if (nv.first == k) { nv.second = v; found = true; }
            if (!found) d.vars.emplace_back(k, v);
        } else if (section == "camera") d.cameras.back()[k] = v;
        else (section == "general" ? d.general : section == "all" ? d.all : section == "options" ? d.options : d.output)[k] = v;
    }
    auto &o = d.options;                                              // defaults: cpp/lib/upsp_inputs.cpp:29-33
    o.emplace("target_patcher", "none");
    o.emplace("registration", "none");
    o.emplace("filter", "none");
    o.emplace("filter_size", "1");
    o.emplace("overlap", "average_view");
    o.emplace("oblique_angle", "70");
    // validation: cpp/exec/psp_process.cpp:1284-1319
    if (d.general.count("tunnel") && d.general["tunnel"] != "ames_unitary") throw DeckError("only tunnel = ames_unitary is supported");
    if (o["registration"] != "none" && o["registration"] != "pixel") throw DeckError("registration must be none or pixel");
    if (o["filter"] != "none" && o["filter"] != "gaussian" && o["filter"] != "box") throw DeckError("filter must be none, gaussian or box");
    if (o["filter"] != "none" && std::atoi(o["filter_size"].c_str()) % 2 == 0) throw DeckError("filter_size must be odd");
    if (o["overlap"] != "best_view" && o["overlap"] != "average_view") throw DeckError("overlap must be best_view or average_view");
    if (o["target_patcher"] != "none")
        throw DeckError("target_patcher = " + o["target_patcher"] + ": phase 0 (patch set-up) is served by the Python driver, not by psp_process_cpp");
    if (d.cameras.empty()) throw DeckError("no @camera section");
    return d;
}

// ---- Cart3D unformatted .tri (cpp/lib/TriModel.ipp:117-257) --------------------------------------------------------------
struct TriGrid {
    std::vector<float> xyz;       // [N][3]
    std::vector<int32_t> tris;    // [T][3], 0-based
};

TriGrid read_tri(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw DeckError("cannot open grid " + path);
    std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t off = 0;
    auto record = [&](size_t &n) -> const char * {
        int32_t head = 0, tail = 0;
        if (off + 4 > buf.size()) throw DeckError("corrupt Fortran record in " + path);
        std::memcpy(&head, &buf[off], 4);
        if (head < 0 || off + 8 + (size_t)head > buf.size()) throw DeckError("corrupt Fortran record in " + path);
        std::memcpy(&tail, &buf[off + 4 + head], 4);
        if (tail != head) throw DeckError("corrupt Fortran record in " + path);
        const char *body = &buf[off + 4];
        off += 8 + (size_t)head;
        n = (size_t)head;
        return body;
    };
    size_t n = 0;
    const char *r = record(n);
    int32_t nn = 0, nt = 0;
    if (n < 8) throw DeckError("corrupt header in " + path);
    std::memcpy(&nn, r, 4);
    std::memcpy(&nt, r + 4, 4);
    TriGrid g;
    r = record(n);
    if (n != (size_t)nn * 12) throw DeckError("node record of " + path + " has the wrong size");
    g.xyz.resize((size_t)nn * 3);
    std::memcpy(g.xyz.data(), r, n);
    r = record(n);
    if (n != (size_t)nt * 12) throw DeckError("triangle record of " + path + " has the wrong size");
    g.tris.resize((size_t)nt * 3);
    std::memcpy(g.tris.data(), r, n);
    for (auto &v : g.tris) v -= 1;                                   // 1-based in the file
    return g;
}

// TriModel_::calcNormals (cpp/lib/TriModel.ipp:1429-1506): normalised sum of the unit face normals (n2 - n1) x (n0 - n1), float
// arithmetic, the magnitudes through double -- operation for operation what upsp_processing_amd/synthetic.py::node_normals does,
// so that both drivers hand the same bits to the oblique-angle test and to the weights.
std::vector<float> node_normals(const TriGrid &g)
{
    const size_t N = g.xyz.size() / 3, T = g.tris.size() / 3;
    std::vector<float> fn(3 * T), acc(3 * N, 0.f);
    for (size_t t = 0; t < T; ++t) {
        const float *n0 = &g.xyz[3 * (size_t)g.tris[3 * t]], *n1 = &g.xyz[3 * (size_t)g.tris[3 * t + 1]], *n2 = &g.xyz[3 * (size_t)g.tris[3 * t + 2]];
        const float a[3] = {n2[0] - n1[0], n2[1] - n1[1], n2[2] - n1[2]}, b[3] = {n0[0] - n1[0], n0[1] - n1[1], n0[2] - n1[2]};
        float c[3];
        {   // numpy.cross: products first, then the difference, each rounded to float
            const float p0 = a[1] * b[2], p1 = a[2] * b[1], p2 = a[2] * b[0], p3 = a[0] * b[2], p4 = a[0] * b[1], p5 = a[1] * b[0];
            c[0] = p0 - p1;
            c[1] = p2 - p3;
            c[2] = p4 - p5;
        }
        const double m2 = ((double)c[0] * (double)c[0] + (double)c[1] * (double)c[1]) + (double)c[2] * (double)c[2];
        const float mag = (float)std::sqrt(m2);
        for (int k = 0; k < 3; ++k) fn[3 * t + k] = mag == 0.f ? c[k] : c[k] / mag;
    }
    for (int k = 0; k < 3; ++k)                                      // np.add.at per corner, triangles in order
        for (size_t t = 0; t < T; ++t) {
            float *dst = &acc[3 * (size_t)g.tris[3 * t + k]];
            dst[0] += fn[3 * t];
            dst[1] += fn[3 * t + 1];
            dst[2] += fn[3 * t + 2];
        }
    for (size_t n = 0; n < N; ++n) {
        float *v = &acc[3 * n];
        const double m2 = ((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1]) + (double)v[2] * (double)v[2];
        const float mag = (float)std::sqrt(m2);
        if (mag != 0.f)
            for (int k = 0; k < 3; ++k) v[k] = v[k] / mag;
    }
    return acc;
}

// ---- camera calibration JSON (read_json_camera_calibration, cpp/lib/CameraCal.cpp:19-54) -----------------------------------
// The numbers under a key, flattened in file order (arrays of any nesting).
std::vector<double> json_numbers(const std::string &txt, const std::string &key, const std::string &path)
{
    const size_t k = txt.find("\"" + key + "\"");
    if (k == std::string::npos) throw DeckError("camera calibration " + path + ": no \"" + key + "\"");
    size_t p = txt.find(':', k);
    if (p == std::string::npos) throw DeckError("camera calibration " + path + ": malformed \"" + key + "\"");
    ++p;
    std::vector<double> out;
    int depth = 0;
    for (; p < txt.size(); ++p) {
        const char c = txt[p];
        if (c == '[') ++depth;
        else if (c == ']') { if (--depth <= 0) break; }
        else if (c == '-' || c == '+' || c == '.' || (c >= '0' && c <= '9')) {
            char *end = nullptr;
            out.push_back(std::strtod(&txt[p], &end));
            p = (size_t)(end - txt.data()) - 1;
            if (depth == 0) break;
        } else if (depth == 0 && (c == ',' || c == '}')) break;
    }
    return out;
}

upsp_camera read_camera(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw DeckError("cannot open camera calibration " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string txt = ss.str();
    upsp_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    const auto K = json_numbers(txt, "cameraMatrix", path), dc = json_numbers(txt, "distCoeffs", path), R = json_numbers(txt, "rmat", path),
               t = json_numbers(txt, "tvec", path), sz = json_numbers(txt, "imageSize", path);
    if (K.size() != 9 || R.size() != 9 || t.size() != 3 || sz.size() != 2) throw DeckError("camera calibration " + path + ": unexpected array sizes");
    std::copy(K.begin(), K.end(), cam.K);
    for (size_t i = 0; i < std::min<size_t>(4, dc.size()); ++i) cam.dist[i] = dc[i];      // (only the first four are read)
    std::copy(R.begin(), R.end(), cam.R);
    std::copy(t.begin(), t.end(), cam.t);
    cam.width = (int)sz[0];
    cam.height = (int)sz[1];
    return cam;
}

// ---- Photron MRAW (cpp/lib/MrawReader.cpp): properties from the .cih header, 12-bit packed frames ---------------------------
struct Mraw {
    std::string path;
    int width = 0, height = 0, bit_depth = 0, num_frames = 0;
    size_t frame_bytes = 0;
    int fd = -1;
};

Mraw open_mraw(const std::string &path)
{
    Mraw m;
    m.path = path;
    const std::string cih = path.substr(0, path.rfind('.')) + ".cih";
    std::ifstream in(cih);
    if (!in) throw DeckError("Video File is invalid (no " + cih + ")");
    std::string line;
    std::map<std::string, std::string> tok;
    while (std::getline(in, line)) {
        line = trim(line);
        // TOKEN_DELIMITER "\s:\s" (MrawReader.cpp:79): exactly two fields
        std::vector<std::string> parts;
        size_t start = 0;
        for (size_t p = 1; p + 1 < line.size(); ++p)
            if (line[p] == ':' && std::isspace((unsigned char)line[p - 1]) && std::isspace((unsigned char)line[p + 1])) {
                parts.push_back(line.substr(start, p - 1 - start));
                start = p + 2;
                ++p;
            }
        parts.push_back(line.substr(start));
        if (parts.size() == 2) tok[parts[0]] = parts[1];
    }
    auto num = [&](const char *k) {
        if (!tok.count(k)) throw DeckError(std::string("cih header: no '") + k + "'");
        return std::atoi(tok[k].c_str());
    };
    m.width = num("Image Width");
    m.height = num("Image Height");
    m.bit_depth = num("Color Bit");
    m.num_frames = num("Total Frame");
    if (m.bit_depth != 12) throw DeckError("only 12-bit MRAW is supported (like the reference)");
    m.frame_bytes = (size_t)m.width * m.height * 12 / 8;
    m.fd = ::open(path.c_str(), O_RDONLY);
    if (m.fd < 0) throw DeckError("Video File is invalid");
    return m;
}

void read_packed(const Mraw &m, int64_t first0, int count, uint8_t *dst)      // frames first0 .. (0-based)
{
    size_t want = (size_t)count * m.frame_bytes, got = 0;
    while (got < want) {
        const ssize_t r = ::pread(m.fd, dst + got, want - got, (off_t)((size_t)first0 * m.frame_bytes + got));
        if (r <= 0) throw std::runtime_error("short read from " + m.path);
        got += (size_t)r;
    }
}

// ---- small helpers -----------------------------------------------------------------------------------------------------
void apportion(int64_t value, int nbins, std::vector<int64_t> &start, std::vector<int64_t> &extent)      // psp_process.cpp:611-624
{
    start.assign(nbins, 0);
    extent.assign(nbins, 0);
    int64_t next = 0;
    for (int b = 0; b < nbins; ++b) {
        start[b] = next;
        extent[b] = value / nbins + (b < value % nbins ? 1 : 0);
        next += extent[b];
    }
}

// The cuts of a rank's frames into K exchange chunks (exchange.hip aligned_chunks: boundaries on multiples of 64 frames, round half
// to even) and the smallest K that keeps every chunk of every rank within `limit` frames (distributed.chunk_count)
std::vector<int64_t> aligned_chunk_extents(int64_t nframes, int K)
{
    std::vector<int64_t> b(K + 1);
    for (int k = 0; k < K; ++k) b[k] = std::min<int64_t>(nframes, (int64_t)std::nearbyint((double)k * (double)nframes / (double)K / 64.0) * 64);
    b[K] = nframes;
    for (int k = 1; k <= K; ++k) b[k] = std::max(b[k], b[k - 1]);
    std::vector<int64_t> e(K);
    for (int k = 0; k < K; ++k) e[k] = b[k + 1] - b[k];
    return e;
}
int chunk_count(const std::vector<int64_t> &frame_counts, int64_t limit)
{
    const int64_t nmax = *std::max_element(frame_counts.begin(), frame_counts.end());
    int K = (int)std::max<int64_t>(1, (nmax + limit - 1) / limit);
    for (;; ++K) {
        int64_t worst = 0;
        for (int64_t n : frame_counts)
            for (int64_t e : aligned_chunk_extents(n, K)) worst = std::max(worst, e);
        if (worst <= limit) return K;
    }
}

void write_file(const std::string &path, const void *p, size_t bytes)
{
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, 1, bytes, f) != bytes) throw std::runtime_error("cannot write " + path);
    std::fclose(f);
}

// regression slices vv-*.dat (psp_process.cpp:1984-2016)
void dump_vv(const std::string &path, const std::vector<float> &v, size_t maxels = 1000)
{
    const size_t step = v.size() < maxels ? 1 : v.size() / maxels;
    std::vector<float> s;
    for (size_t i = 0; i < v.size() && s.size() < maxels; i += step) s.push_back(v[i]);
    write_file(path, s.data(), s.size() * sizeof(float));
}

uint32_t crc32_of(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

// camNN-nodecount.png (psp_process.cpp:1608-1614): the nodes-per-pixel image through upsp::nodes_per_pixel_colormap
// (cpp/utils/cv_extras.cpp:277-290); 8-bit RGB PNG, the zlib stream in stored (uncompressed) blocks.
void write_nodecount_png(const std::string &path, const std::vector<uint8_t> &counts, int w, int h)
{
    static const uint8_t lut[5][3] = {{0, 0, 0}, {0, 255, 0}, {255, 255, 0}, {255, 153, 51}, {255, 204, 153}};
    std::vector<uint8_t> raw((size_t)h * (1 + 3 * (size_t)w));
    for (int y = 0; y < h; ++y) {
        uint8_t *row = &raw[(size_t)y * (1 + 3 * (size_t)w)];
        row[0] = 0;                                                   // filter type 0
        for (int x = 0; x < w; ++x) {
            const uint8_t c = counts[(size_t)y * w + x];
            for (int k = 0; k < 3; ++k) row[1 + 3 * x + k] = c < 5 ? lut[c][k] : 255;
        }
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;                                            // adler32
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        if (raw.empty()) break;
    }
    const uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; --k) z.push_back((uint8_t)(adler >> (8 * k)));
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    auto chunk = [&](const char *tag, const std::vector<uint8_t> &data) {
        std::vector<uint8_t> td(tag, tag + 4);
        td.insert(td.end(), data.begin(), data.end());
        const uint32_t len = (uint32_t)data.size(), crc = crc32_of(td.data(), td.size());
        for (int k = 3; k >= 0; --k) png.push_back((uint8_t)(len >> (8 * k)));
        png.insert(png.end(), td.begin(), td.end());
        for (int k = 3; k >= 0; --k) png.push_back((uint8_t)(crc >> (8 * k)));
    };
    std::vector<uint8_t> ihdr;
    for (uint32_t v : {(uint32_t)w, (uint32_t)h})
        for (int k = 3; k >= 0; --k) ihdr.push_back((uint8_t)(v >> (8 * k)));
    for (uint8_t v : {8, 2, 0, 0, 0}) ihdr.push_back(v);
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    write_file(path, png.data(), png.size());
}

template <typename T>
T *dev_alloc(size_t n)
{
    T *p = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&p), sizeof(T) * std::max<size_t>(n, 1)));
    return p;
}
template <typename T>
T *to_device(const std::vector<T> &h)
{
    T *d = dev_alloc<T>(h.size());
    if (!h.empty()) HIPCHECK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return d;
}
template <typename T>
std::vector<T> to_host(const T *d, size_t n)
{
    std::vector<T> h(n);
    if (n) HIPCHECK(hipMemcpy(h.data(), d, sizeof(T) * n, hipMemcpyDeviceToHost));
    return h;
}

int64_t series_ld(int64_t nframes)           // engine.series_ld: rows on 256-byte boundaries, an odd multiple for column-chunk fills
{
    const int64_t ld = (nframes + 63) / 64 * 64;
    return (ld / 64) % 2 == 0 ? ld + 64 : ld;
}

// deg2_rad(180 - oblique_angle) narrowed to float (psp_process.cpp:1602; engine.oblique_threshold)
float oblique_threshold(double angle_deg)
{
    const float a = (float)angle_deg;
    return (float)((180.0 - (double)a) * 3.141592653589793 / 180.0);
}

// ---- the run of one rank --------------------------------------------------------------------------------------------
int run_rank(const std::map<std::string, std::string> &flags, int rank, int world, const std::string &id_file)
{
    auto flag = [&](const char *k) { auto it = flags.find(k); return it == flags.end() ? std::string() : it->second; };
    Deck deck = parse_deck(flag("input_file"));
    auto &opts = deck.options;
    const std::string grid = deck.all.count("grid") ? deck.all["grid"] : std::string();
    if (!ends_with(grid, ".tri")) throw DeckError("psp_process_cpp reads Cart3D .tri grids (got '" + grid + "'; PLOT3D grids: the Python driver)");
    const TriGrid g = read_tri(grid);
    const size_t N = g.xyz.size() / 3, T = g.tris.size() / 3;
    const std::vector<float> normals = node_normals(g);
    std::vector<float> soup(9 * T);                                  // TriModel_::extract_tris (cpp/lib/TriModel.ipp:261-299)
    for (size_t k = 0; k < 3 * T; ++k) std::memcpy(&soup[3 * k], &g.xyz[3 * (size_t)g.tris[k]], 3 * sizeof(float));
    std::vector<uint8_t> datanode;
    if (!flag("cutoff_x_max").empty()) {                              // psp_process.cpp:1448-1487
        const float cut = std::strtof(flag("cutoff_x_max").c_str(), nullptr);
        datanode.resize(N);
        for (size_t n = 0; n < N; ++n) datanode[n] = g.xyz[3 * n] <= cut ? 1 : 0;
    }
    const int C = (int)deck.cameras.size();
    if (C > 16) throw DeckError("too many cameras (the pipeline holds at most 16)");
    std::vector<upsp_camera> cams;
    std::vector<Mraw> videos;
    for (auto &c : deck.cameras) {
        if (!c.count("calibration")) throw DeckError("@camera without calibration");
        cams.push_back(read_camera(c["calibration"]));
        const std::string fn = c.count("filename") ? c["filename"] : c.count("cine") ? c["cine"] : std::string();
        if (!ends_with(fn, ".mraw")) throw DeckError("psp_process_cpp reads .mraw videos (got '" + fn + "'; CINE files: the Python driver)");
        videos.push_back(open_mraw(fn));
    }
    const int W = cams[0].width, H = cams[0].height;
    for (int c = 0; c < C; ++c)
        if (cams[c].width != W || cams[c].height != H || videos[c].width != W || videos[c].height != H)
            throw DeckError("camera calibration / video sizes disagree");
    int64_t nframes = videos[0].num_frames;
    for (auto &v : videos) nframes = std::min<int64_t>(nframes, v.num_frames);
    if (opts.count("number_frames") && std::atoi(opts["number_frames"].c_str()) > 0) nframes = std::min<int64_t>(nframes, std::atoi(opts["number_frames"].c_str()));
    if (!flag("frames").empty() && std::atoi(flag("frames").c_str()) > 0) nframes = std::min<int64_t>(nframes, std::atoi(flag("frames").c_str()));
    const size_t npix = (size_t)W * H;

    // ---- communicator (MPI_Init / MPI_Comm_rank of the reference, :1322-1330) ----
    upsp_comm *comm = nullptr;
    if (world > 1) {
        uint8_t id[128];
        if (rank == 0) {
            CHECK(upsp_comm_unique_id(id));
            // (O_EXCL in the launcher's private directory: nobody else's file or link is ever written through)
            const std::string tmp = id_file + ".tmp";
            const int fd = ::open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
            if (fd < 0 || ::write(fd, id, 128) != 128) throw std::runtime_error("cannot publish the communicator id in " + tmp);
            ::close(fd);
            if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish the communicator id");
        } else {
            std::FILE *f = nullptr;
            for (int i = 0; i < 1200 && !(f = std::fopen(id_file.c_str(), "rb")); ++i) usleep(100000);
            if (!f || std::fread(id, 1, 128, f) != 128) throw std::runtime_error("rank 0 never published the communicator id");
            std::fclose(f);
        }
        CHECK(upsp_comm_create(id, rank, world, &comm));
    }
    std::vector<int64_t> fstart, fcount, nstart, ncount;
    apportion(nframes, world, fstart, fcount);                        // rank_start_frame / rank_num_frames, :1519-1529
    apportion((int64_t)N, world, nstart, ncount);
    const int64_t f0 = fstart[rank], nf = fcount[rank];

    // ---- phase 1: BVH + projection matrices ----
    upsp_bvh *bvh = nullptr;
    CHECK(upsp_bvh_create(soup.data(), T, &bvh));                     // createBVH :44-53
    float *d_nodes = to_device(g.xyz), *d_normals = to_device(normals);
    int32_t *d_tri_nodes = to_device(g.tris);
    uint8_t *d_datanode = datanode.empty() ? nullptr : to_device(datanode);
    CHECK(upsp_bvh_set_tri_nodes(bvh, d_tri_nodes, N, nullptr));
    const float thresh = oblique_threshold(std::strtod(opts["oblique_angle"].c_str(), nullptr));
    int32_t *d_pix = dev_alloc<int32_t>((size_t)C * N);               // [C][N]
    std::vector<float *> d_uv(C);
    std::vector<uint8_t *> d_cnt(C);
    uint64_t rays_cast = 0;
    std::vector<double> centers(3 * (size_t)C);
    for (int c = 0; c < C; ++c) {                                     // :1597-1622
        d_uv[c] = dev_alloc<float>(2 * N);
        d_cnt[c] = dev_alloc<uint8_t>(npix);
        CHECK(upsp_projection_build(bvh, &cams[c], d_nodes, d_normals, d_datanode, d_tri_nodes, N, thresh, d_pix + (size_t)c * N, d_uv[c],
                                    d_cnt[c], nullptr, nullptr));
        uint64_t nr = 0, pr = 0, rn = 0;
        CHECK(upsp_projection_fetch_counts(bvh, &nr, &pr, &rn, nullptr));
        rays_cast += nr;
        CHECK(upsp_camera_center(&cams[c], &centers[3 * (size_t)c]));
    }
    float *d_weight = nullptr;                                        // :1632-1640; a single camera needs no weights
    if (C > 1) {
        std::vector<float> ones((size_t)C * N, 1.0f);
        d_weight = to_device(ones);
        CHECK(upsp_projection_weights(C, N, d_pix, d_weight, d_nodes, d_normals, centers.data(), opts["overlap"] == "average_view" ? 1 : 0, nullptr));
    }
    uint8_t *d_skipped = dev_alloc<uint8_t>(N);
    CHECK(upsp_projection_skipped(C, N, d_pix, d_skipped, nullptr, nullptr));      // :1644
    upsp_pipeline_opts po;
    upsp_pipeline_default_opts(&po);
    po.registration = opts["registration"] == "pixel" ? 1 : 0;
    po.filter = opts["filter"] == "gaussian" ? 1 : opts["filter"] == "box" ? 2 : 0;
    if (po.filter) po.filter_size = std::atoi(opts["filter_size"].c_str());
    upsp_pipeline *pipe = nullptr;
    CHECK(upsp_pipeline_create(C, W, H, N, &po, &pipe));
    for (int c = 0; c < C; ++c) CHECK(upsp_pipeline_set_projection(pipe, c, d_pix + (size_t)c * N, d_weight ? d_weight + (size_t)c * N : nullptr));
    CHECK(upsp_pipeline_set_skipped(pipe, d_skipped));

    // ---- first frame (psp_process.cpp:1655-1713): ECC template = the raw first frame as f32; sol1 = the repaired first frame
    //      registered against its own f32 copy, filtered, projected ----
    std::vector<uint16_t *> d_first(C);
    std::vector<float *> d_ref(C), d_ref_fixed(C);
    std::vector<uint8_t> packed1(videos[0].frame_bytes);
    uint8_t *d_packed1 = dev_alloc<uint8_t>(videos[0].frame_bytes);
    for (int c = 0; c < C; ++c) {
        read_packed(videos[c], 0, 1, packed1.data());
        HIPCHECK(hipMemcpy(d_packed1, packed1.data(), packed1.size(), hipMemcpyHostToDevice));
        d_first[c] = dev_alloc<uint16_t>(npix);
        CHECK(upsp_unpack_12bit(d_packed1, 1, npix, d_first[c], 0, nullptr, nullptr));
        std::vector<uint16_t> raw = to_host(d_first[c], npix);
        std::vector<float> r32(raw.begin(), raw.end());
        d_ref[c] = to_device(r32);
        CHECK(upsp_pipeline_set_reference(pipe, c, d_ref[c]));
        CHECK(upsp_fix_hot_pixels(d_first[c], 1, H, W, po.hot_thresh, po.hot_min_change, po.hot_max, nullptr, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        std::vector<uint16_t> fixed = to_host(d_first[c], npix);
        std::vector<float> f32(fixed.begin(), fixed.end());
        d_ref_fixed[c] = to_device(f32);
    }
    std::vector<float> sol1(N);
    {
        upsp_pipeline_opts p1 = po;
        p1.hot_enable = 0;
        upsp_pipeline *tmp = nullptr;
        CHECK(upsp_pipeline_create(C, W, H, N, &p1, &tmp));
        for (int c = 0; c < C; ++c) {
            CHECK(upsp_pipeline_set_projection(tmp, c, d_pix + (size_t)c * N, d_weight ? d_weight + (size_t)c * N : nullptr));
            CHECK(upsp_pipeline_set_reference(tmp, c, d_ref_fixed[c]));
        }
        CHECK(upsp_pipeline_set_skipped(tmp, d_skipped));
        float *d_rows1 = dev_alloc<float>(N);
        CHECK(upsp_pipeline_process(tmp, d_first.data(), 1, 1, d_rows1, nullptr, 0, 0, nullptr, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        sol1 = to_host(d_rows1, N);
        upsp_pipeline_destroy(tmp);
        (void)hipFree(d_rows1);
    }

    // ---- frame loop (psp_process.cpp:1743-1851): chunks of <= 256 frames, disk -> pinned slot -> device -> unpack -> process ----
    const int64_t ld = series_ld(std::max<int64_t>(nf, 1));
    float *d_rows_t = dev_alloc<float>(N * (size_t)ld);
    CHECK(upsp_pipeline_set_row_padding(pipe, 1));      // (columns nf .. ld of that allocation are padding)
    const int chunk = 256;
    std::vector<uint8_t *> h_pinned(C), d_packed(C);
    std::vector<uint16_t *> d_frames(C);
    for (int c = 0; c < C; ++c) {
        HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&h_pinned[c]), (size_t)chunk * videos[c].frame_bytes, hipHostMallocDefault));
        d_packed[c] = dev_alloc<uint8_t>((size_t)chunk * videos[c].frame_bytes);
        d_frames[c] = dev_alloc<uint16_t>((size_t)chunk * npix);
    }
    // N > 1, one camera, no weights, no float image stage: the ACTIVE PIXELS' u16 series travel chunk by chunk while the next chunk
    // is read and scanned, and the owner of a node runs pass B over all frames (psp.Phase1.frame_loop_pixel_wire; a third of the
    // bytes of the node rows on the links).  UPSP_ROW_WIRE=1 forces the node rows.
    const bool pixel_wire = comm && C == 1 && !po.registration && !po.filter && !std::getenv("UPSP_ROW_WIRE");
    upsp_exchange *px = nullptr;
    float *d_series_px = nullptr;
    if (pixel_wire) {
        const int64_t limit = std::min<int64_t>(chunk, upsp_pipeline_series_frames_max(pipe));
        const int K = chunk_count(fcount, limit);
        CHECK(upsp_exchange_create(comm, nframes, (int64_t)N, K, &px));
        const uint16_t *d_compact = nullptr;
        uint32_t cpitch = 0;
        const int32_t *d_node_k = nullptr;
        CHECK(upsp_pipeline_pixel_series(pipe, nullptr, 0, nullptr, &d_compact, &cpitch, &d_node_k, nullptr));     // the node -> row table alone
        CHECK(upsp_exchange_set_pixels(px, d_node_k, d_skipped, 0, nullptr));
        for (int k = 0; k < K; ++k) {
            int64_t c0 = 0, fc = 0;
            CHECK(upsp_exchange_chunk(px, k, &c0, &fc));
            if (fc > 0) {
                HIPCHECK(hipDeviceSynchronize());
                read_packed(videos[0], f0 + c0, (int)fc, h_pinned[0]);
                HIPCHECK(hipMemcpyAsync(d_packed[0], h_pinned[0], (size_t)fc * videos[0].frame_bytes, hipMemcpyHostToDevice, nullptr));
                CHECK(upsp_unpack_12bit(d_packed[0], (int)fc, npix, d_frames[0], 0, nullptr, nullptr));
                CHECK(upsp_pipeline_pixel_series(pipe, d_frames[0], (int)fc, nullptr, &d_compact, &cpitch, &d_node_k, nullptr));
            }
            CHECK(upsp_exchange_submit_pixels(px, d_compact, cpitch, 2, nullptr));     // (a rank with fewer chunks still takes part)
            if (rank == 0 && k % 4 == 0) std::printf("  Rank 0:: processing frame %lld\n", (long long)(f0 + c0));
        }
        double *ps = nullptr, *pss = nullptr;
        CHECK(upsp_pipeline_accumulators(pipe, &ps, &pss));
        d_series_px = dev_alloc<float>((size_t)std::max<int64_t>(ncount[rank], 1) * (size_t)nframes);
        CHECK(upsp_exchange_finish_pixels(px, d_series_px, nframes, ps + nstart[rank], pss + nstart[rank], nullptr));
        HIPCHECK(hipDeviceSynchronize());
    }
    for (int64_t c0 = 0; c0 < (pixel_wire ? 0 : nf); c0 += chunk) {
        const int n = (int)std::min<int64_t>(chunk, nf - c0);
        HIPCHECK(hipDeviceSynchronize());                             // (the previous chunk's kernels still read the buffers)
        for (int c = 0; c < C; ++c) {
            read_packed(videos[c], f0 + c0, n, h_pinned[c]);
            HIPCHECK(hipMemcpyAsync(d_packed[c], h_pinned[c], (size_t)n * videos[c].frame_bytes, hipMemcpyHostToDevice, nullptr));
            CHECK(upsp_unpack_12bit(d_packed[c], n, npix, d_frames[c], 0, nullptr, nullptr));
        }
        CHECK(upsp_pipeline_process(pipe, d_frames.data(), n, f0 + c0, nullptr, d_rows_t, ld, c0, nullptr, nullptr));
        if (rank == 0 && c0 % (chunk * 4) == 0) std::printf("  Rank 0:: processing frame %lld\n", (long long)(f0 + c0));
    }
    HIPCHECK(hipDeviceSynchronize());

    // ---- reductions + finals (psp_process.cpp:1866-1979) ----
    double *d_sum = nullptr, *d_sumsq = nullptr;
    CHECK(upsp_pipeline_accumulators(pipe, &d_sum, &d_sumsq));
    if (comm) CHECK(upsp_allreduce_sums(comm, d_sum, d_sumsq, N, nullptr));          // MPI_Reduce + MPI_Bcast
    float *d_avg = dev_alloc<float>(N), *d_rms = dev_alloc<float>(N);
    CHECK(upsp_pipeline_finalize(pipe, (uint64_t)nframes, d_avg, d_rms, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    const std::vector<float> avg = to_host(d_avg, N), rms = to_host(d_rms, N);
    std::vector<float> ratio0(N), coverage(N, 0.f);
    for (size_t n = 0; n < N; ++n) ratio0[n] = avg[n] / sol1[n] - 1.0f;               // :1948-1950
    {   // coverage = sum_c project(ones) (:1955-1975)
        std::vector<float> ones(npix, 1.0f);
        float *d_ones = to_device(ones), *d_ind = dev_alloc<float>(N);
        for (int c = 0; c < C; ++c) {
            CHECK(upsp_project_frame_f32(d_ones, d_pix + (size_t)c * N, d_weight ? d_weight + (size_t)c * N : nullptr, N, d_ind, nullptr));
            HIPCHECK(hipDeviceSynchronize());
            const std::vector<float> ind = to_host(d_ind, N);
            for (size_t n = 0; n < N; ++n) coverage[n] = c == 0 ? ind[n] : coverage[n] + ind[n];
        }
        (void)hipFree(d_ones);
        (void)hipFree(d_ind);
    }

    // ---- time-series exchange (global_transpose, :707-771): [N][frames of this rank] -> [nodes of this rank][all frames] ----
    const int64_t n0 = nstart[rank], nn = ncount[rank];
    std::vector<float> series((size_t)nn * (size_t)nframes);
    if (pixel_wire) {
        series = to_host(d_series_px, (size_t)nn * (size_t)nframes);
        upsp_exchange_destroy(px);
        (void)hipFree(d_series_px);
    } else if (!comm) {
        std::vector<float> rows = to_host(d_rows_t, N * (size_t)ld);
        for (size_t n = 0; n < N; ++n) std::memcpy(&series[n * (size_t)nframes], &rows[n * (size_t)ld], sizeof(float) * (size_t)nframes);
    } else {
        upsp_exchange *x = nullptr;
        CHECK(upsp_exchange_create(comm, nframes, (int64_t)N, 1, &x));
        CHECK(upsp_exchange_set_skipped(x, nullptr, 0, nullptr));     // every row travels (cameras, weights and float stages allowed)
        float *d_send = dev_alloc<float>(N * (size_t)std::max<int64_t>(nf, 1));
        if (nf > 0)
            HIPCHECK(hipMemcpy2D(d_send, sizeof(float) * (size_t)nf, d_rows_t, sizeof(float) * (size_t)ld, sizeof(float) * (size_t)nf, N, hipMemcpyDeviceToDevice));
        CHECK(upsp_exchange_submit(x, d_send, 4, nullptr));
        float *d_series = dev_alloc<float>((size_t)std::max<int64_t>(nn, 1) * (size_t)nframes);
        CHECK(upsp_exchange_finish(x, d_series, nframes, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        series = to_host(d_series, (size_t)nn * (size_t)nframes);
        upsp_exchange_destroy(x);
        (void)hipFree(d_send);
        (void)hipFree(d_series);
    }

    // ---- flat files (:524-540; docs/sphinx/file-formats.rst:821-894): raw little-endian f32, no header ----
    std::string out_dir = !flag("add_out_dir").empty() ? flag("add_out_dir") : deck.output.count("dir") ? deck.output["dir"] : std::string(".");
    ::mkdir(out_dir.c_str(), 0777);
    auto out = [&](const std::string &name) { return out_dir + "/" + name; };
    if (rank == 0) {
        write_file(out("intensity_avg"), avg.data(), N * 4);
        write_file(out("intensity_rms"), rms.data(), N * 4);
        write_file(out("intensity_ratio_0"), ratio0.data(), N * 4);
        write_file(out("coverage"), coverage.data(), N * 4);
        for (int c = 0; c < C; ++c) {
            char name[64];
            std::snprintf(name, sizeof(name), "cam%02d-uv", c + 1);
            const std::vector<float> uv = to_host(d_uv[c], 2 * N);
            write_file(out(name), uv.data(), uv.size() * 4);
            std::snprintf(name, sizeof(name), "cam%02d-nodecount.png", c + 1);
            write_nodecount_png(out(name), to_host(d_cnt[c], npix), W, H);
        }
        dump_vv(out("vv-int-rms.dat"), rms);
        dump_vv(out("vv-int-avg.dat"), avg);
        dump_vv(out("vv-int-coverage.dat"), coverage);
        dump_vv(out("vv-int-sample1.dat"), ratio0);
        const char *axes[3] = {"X", "Y", "Z"};
        for (int a = 0; a < 3; ++a) {
            std::vector<float> col(N);
            for (size_t n = 0; n < N; ++n) col[n] = g.xyz[3 * n + a];
            write_file(out(axes[a]), col.data(), N * 4);
        }
        // rank 0 creates / truncates the shared file every rank then writes its node slice into (:958-963)
        const int fd = ::open(out("intensity_transpose").c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
        if (fd < 0 || ::ftruncate(fd, (off_t)(N * (size_t)nframes * 4)) != 0) throw std::runtime_error("cannot create intensity_transpose");
        ::close(fd);
    }
    if (comm) {      // barrier: nobody writes into a file that is about to be truncated (an all-reduce of one element per array)
        double *d_b = dev_alloc<double>(2);
        HIPCHECK(hipMemset(d_b, 0, 16));
        CHECK(upsp_allreduce_sums(comm, d_b, d_b + 1, 1, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        (void)hipFree(d_b);
    }
    {
        const int fd = ::open(out("intensity_transpose").c_str(), O_WRONLY);
        if (fd < 0) throw std::runtime_error("cannot open intensity_transpose");
        const size_t bytes = series.size() * 4;
        size_t done = 0;
        while (done < bytes) {
            const ssize_t w = ::pwrite(fd, reinterpret_cast<const char *>(series.data()) + done, bytes - done, (off_t)((size_t)n0 * (size_t)nframes * 4 + done));
            if (w <= 0) throw std::runtime_error("short write to intensity_transpose");
            done += (size_t)w;
        }
        ::close(fd);
    }
    if (rank == 0) std::printf("phase 1 complete: %lld frames, %zu nodes, %llu rays cast\n", (long long)nframes, N, (unsigned long long)rays_cast);
    upsp_pipeline_destroy(pipe);
    upsp_bvh_destroy(bvh);
    if (comm) {
        HIPCHECK(hipDeviceSynchronize());
        upsp_comm_destroy(comm);
    }
    return 0;
}

// `-ranks=N`: N rank processes (the `mpiexec -n N psp_process` of the reference's batch templates), started before this process
// has touched a GPU; rank r runs on device r (UPSP_ONE_GPU=1: all on device 0 -- the tests, with UPSP_RCCL_LIBRARY naming an
// RCCL that allows it).  Like mpiexec, the launcher tears the job down when one rank fails: the first rank that ends with a
// non-zero status (or by a signal) gets its peers -- which would otherwise wait for it in upsp_allreduce_sums / upsp_exchange_*
// for ever -- a SIGTERM and, after a grace period, a SIGKILL; the exit code is that first failure's.
// NEVER under rocprofv3: the profiler's preloaded tool has initialised the GPU before main(), and fork + exec from such a process
// is what the pool forbids.  Profile a rank by starting the N processes yourself with UPSP_RANK / UPSP_WORLD / UPSP_ID_FILE set
// (INTEGRATION.md).  The communicator id travels through a file in a private 0700 directory (mkdtemp), written with O_EXCL.
int spawn_ranks(int n, int argc, char **argv)
{
    (void)argc;
    char dir[] = "/tmp/upsp_psp_XXXXXX";
    if (!mkdtemp(dir)) {
        std::perror("mkdtemp");
        return 2;
    }
    const std::string id_file = std::string(dir) + "/id";
    std::vector<pid_t> kids;
    for (int r = 0; r < n; ++r) {
        const pid_t p = fork();
        if (p == 0) {
            setenv("UPSP_RANK", std::to_string(r).c_str(), 1);
            setenv("UPSP_WORLD", std::to_string(n).c_str(), 1);
            setenv("UPSP_ID_FILE", id_file.c_str(), 1);
            execv("/proc/self/exe", argv);
            std::perror("execv");
            _exit(127);
        }
        if (p < 0) {
            std::perror("fork");
            break;
        }
        kids.push_back(p);
    }
    int rc = (int)kids.size() == n ? 0 : 2;
    size_t left = kids.size();
    bool killing = rc != 0;
    auto signal_all = [&](int sig) {
        for (pid_t k : kids)
            if (k > 0) ::kill(k, sig);
    };
    if (killing) signal_all(SIGTERM);
    int grace = 0;                                   // tenths of a second since the SIGTERM
    while (left > 0) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, killing ? WNOHANG : 0);
        if (p == 0) {                                // tearing down: poll, SIGKILL after 5 s
            usleep(100000);
            if (++grace == 50) signal_all(SIGKILL);
            continue;
        }
        if (p < 0) {
            if (errno == EINTR) continue;
            break;
        }
        bool mine = false;
        for (pid_t &k : kids)
            if (k == p) {
                k = -1;
                mine = true;
            }
        if (!mine) continue;
        --left;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (code && !killing) {                      // the first failure: its code, and the peers go down with it
            rc = code;
            killing = true;
            std::fprintf(stderr, "psp_process: a rank ended with status %d: stopping the other %zu\n", code, left);
            signal_all(SIGTERM);
        }
    }
    ::unlink(id_file.c_str());
    ::unlink((id_file + ".tmp").c_str());
    ::rmdir(dir);
    return rc;
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        const auto flags = parse_flags(argc, argv);
        const char *er = std::getenv("UPSP_RANK");
        int world = 1, rank = 0;
        std::string id_file;
        if (er) {
            rank = std::atoi(er);
            world = std::atoi(std::getenv("UPSP_WORLD") ? std::getenv("UPSP_WORLD") : "1");
            id_file = std::getenv("UPSP_ID_FILE") ? std::getenv("UPSP_ID_FILE") : "";
        } else if (flags.count("ranks") && std::atoi(flags.at("ranks").c_str()) > 1) {
            return spawn_ranks(std::atoi(flags.at("ranks").c_str()), argc, argv);
        }
        if (world > 1 || er) {
            const int dev = std::getenv("UPSP_ONE_GPU") ? 0 : rank;
            HIPCHECK(hipSetDevice(dev));
        }
        return run_rank(flags, rank, world, id_file);
    } catch (const DeckError &e) {
        std::fprintf(stderr, "psp_process: %s\n", e.what());
        return 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "psp_process: error: %s\n", e.what());
        return 2;
    }
}
