// agg_jit.hip — see agg_jit.hpp.
#include "agg_jit.hpp"

#include <dlfcn.h>

#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <cerrno>
#include <fcntl.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <map>
#include <vector>
#include <thread>
#include <chrono>
#include <atomic>
#include <mutex>
#include <sstream>
#include <string>

extern char **environ;

namespace qsx {

struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t function = nullptr;
};

namespace {

const char *const kBundle =
#include "jit_bundle.inc"
    ;

// hipRTC has the HIP device built-ins but no libc headers.
const char *const kPrelude = R"(
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
typedef unsigned long size_t;
typedef unsigned long uintptr_t;
)";

void emit_operand(std::ostringstream &o, const char *lhs, const DevOperand &v) {
  o << "  " << lhs << ".kind = " << v.kind << "; " << lhs << ".index = " << v.index << ";\n";
}

// The translated configuration as the body of a constexpr function (non-zero fields only; doubles as
// hex-float literals so that they are bit-exact).
std::string emit_dev_config(const DevConfig &d) {
  std::ostringstream o;
  char buf[64];
  o << "  d.num_columns = " << d.num_columns << ";\n";
  for (int i = 0; i < d.num_columns; ++i) {
    o << "  d.column_type[" << i << "] = " << d.column_type[i] << "; d.column_width[" << i << "] = " << d.column_width[i]
      << "; d.lds_off[" << i << "] = " << d.lds_off[i] << ";\n";
    o << "  d.code_off[" << i << "] = " << d.code_off[i] << ";\n";
    if (d.code_width[i] != 0) o << "  d.code_width[" << i << "] = " << d.code_width[i] << ";\n";
  }
  for (int i = d.num_columns; i < QSX_MAX_COLUMNS; ++i) o << "  d.lds_off[" << i << "] = -1; d.code_off[" << i << "] = -1;\n";
  o << "  d.num_keys = " << d.num_keys << ";\n";
  for (int k = 0; k < d.num_keys; ++k) {
    o << "  d.key_column[" << k << "] = " << d.key_column[k] << "; d.key_width[" << k << "] = " << d.key_width[k]
      << "; d.key_shift[" << k << "] = " << d.key_shift[k] << ";\n";
  }
  if (d.wide_words != 0) {
    o << "  d.wide_words = " << d.wide_words << "; d.wide_hash_mask = " << d.wide_hash_mask << "ull;\n";
    for (int k = 0; k < d.num_keys; ++k) o << "  d.key_word[" << k << "] = " << d.key_word[k] << ";\n";
  }
  o << "  d.num_instrs = " << d.num_instrs << ";\n";
  for (int k = 0; k < d.num_instrs; ++k) {
    o << "  d.instrs[" << k << "].op = " << d.instrs[k].op << "; d.instrs[" << k << "].dst = " << d.instrs[k].dst << ";\n";
    std::snprintf(buf, sizeof(buf), "d.instrs[%d].a", k);
    emit_operand(o, buf, d.instrs[k].a);
    std::snprintf(buf, sizeof(buf), "d.instrs[%d].b", k);
    emit_operand(o, buf, d.instrs[k].b);
  }
  for (int k = 0; k < QSX_MAX_CONSTS; ++k) {
    if (d.consts[k] == 0.0 && !std::signbit(d.consts[k])) continue;
    if (d.consts[k] != d.consts[k] || d.consts[k] - d.consts[k] != 0.0) {   // NaN / inf have no literal
      unsigned long long bits;
      std::memcpy(&bits, &d.consts[k], 8);
      o << "  d.consts[" << k << "] = __builtin_bit_cast(double, " << bits << "ull);\n";
    } else {
      std::snprintf(buf, sizeof(buf), "%a", d.consts[k]);
      o << "  d.consts[" << k << "] = " << buf << ";\n";
    }
  }
  o << "  d.num_sums = " << d.num_sums << ";\n";
  for (int j = 0; j < d.num_sums; ++j) {
    std::snprintf(buf, sizeof(buf), "d.sums[%d].arg", j);
    emit_operand(o, buf, d.sums[j].arg);
    o << "  d.sums[" << j << "].is_int = " << d.sums[j].is_int << "; d.sums[" << j << "].kind = " << d.sums[j].kind << ";\n";
    if (d.sums[j].null_mask != 0 || d.sums[j].count_valid != 0) {
      o << "  d.sums[" << j << "].null_mask = " << d.sums[j].null_mask << "u; d.sums[" << j << "].count_valid = " << d.sums[j].count_valid << ";\n";
    }
  }
  // nullable columns the plan reads: which, where their null words of a tile are staged, which of them drop the row (the
  // bitmaps themselves are per call: behind the kernel's `nulls` pointer)
  if (d.num_null_cols != 0) {
    o << "  d.num_null_cols = " << d.num_null_cols << "; d.row_null_mask = " << d.row_null_mask << "u;\n";
    for (int sl = 0; sl < d.num_null_cols; ++sl) {
      o << "  d.null_column[" << sl << "] = " << d.null_column[sl] << "; d.null_lds_off[" << sl << "] = " << d.null_lds_off[sl] << ";\n";
    }
  }
  o << "  d.num_pred = " << d.num_pred << ";\n";
  for (int p = 0; p < d.num_pred; ++p) {
    o << "  d.pred[" << p << "].column = " << d.pred[p].column << "; d.pred[" << p << "].op = " << d.pred[p].op << "; d.pred["
      << p << "].literal = " << d.pred[p].literal << "ull;\n";
  }
  o << "  d.filter_lds_off = " << d.filter_lds_off << "; d.tile_bytes = " << d.tile_bytes << ";\n";
  return o.str();
}

std::string compiler_driver();   // (below: who compiles the shapes — empty = hipRTC)

std::string make_source(const DevConfig &dev, int num_sums, bool dense, const JitGeometry &geo) {
  std::ostringstream o;
  bool any_coded = false;   // unused pointers are passed as literals: every live scalar argument costs SGPRs in the tile loop
  for (int i = 0; i < dev.num_columns; ++i) any_coded = any_coded || dev.code_width[i] != 0;
  // a state over nullable columns: the null bitmaps of a call arrive as a device table of num_null_cols pointers (by null
  // slot; a null entry = the block has no NULL in that attribute) — the last of the trailing arguments
  const bool any_nulls = dev.num_null_cols != 0;
  const char *nulls_param = any_nulls ? ", const unsigned long long *const *nulls" : "";
  const char *nulls_arg = any_nulls ? "nulls" : "nullptr";
  o << kPrelude << kBundle << "\nnamespace qsx {\nconstexpr DevConfig jit_make_dev() {\n  DevConfig d{};\n"
    << emit_dev_config(dev) << "  return d;\n}\n";
  if (geo.dir_gids != 0 && dense) {
    // a dense state in LDS (agg_hash_update.hpp, kDense && kDir): the signature of the plain shapes, 1024 threads
    o << "extern \"C\" __global__ __launch_bounds__(" << kDirBlock << ") void qsx_jit_agg(ColumnPointers cols, int64_t n,\n"
      << "    DenseView view, const long long *pieces"
      << (dev.filter_lds_off >= 0 || any_coded || any_nulls ? ", const uint64_t *filter" : "") << (any_coded || any_nulls ? ", const void *const *dicts" : "")
      << nulls_param << ") {\n"
      << "  static constexpr DevConfig D = jit_make_dev();\n"
      << "  (void)cols; (void)pieces;\n"
      << "  agg_hash_update_body<true, true, " << num_sums << ", " << (geo.dir_rows > 1 ? geo.dir_rows : 1) << ", true, " << kDirBlock << ", false, "
      << (geo.runs != 0 ? "true" : "false")
      << ">(D, " << (geo.runs != 0 ? "nullptr" : "cols.p") << ", " << (any_coded ? "dicts" : "nullptr") << ", n, "
      << (dev.filter_lds_off >= 0 ? "filter" : "nullptr") << ", HashTableView{}, view, " << geo.S << ", " << geo.rep_shift << ", " << geo.nbuf
      << ", " << geo.ranges << ", pieces, " << nulls_arg << ", nullptr);\n}\n}  // namespace qsx\n";
    return o.str();
  }
  if (geo.dir_gids != 0) {
    // group-directory variant: the body of agg_dir_update_kernel with the configuration and the geometry as constants
    // (geo.runs: the rows are a run of blocks, the table arrives as `pieces`)
    o << "extern \"C\" __global__ __launch_bounds__(" << kDirBlock << ") void qsx_jit_agg(ColumnPointers cols,\n"
      << "    const void *const *dicts, int64_t n, const uint64_t *filter, HashTableView view, DirView d, const long long *pieces"
      << nulls_param << ") {\n"
      << "  static constexpr DevConfig D = jit_make_dev();\n"
      << "  (void)cols; (void)pieces;\n"
      << "  agg_hash_update_body<true, false, " << num_sums << ", " << (geo.dir_rows == 2 ? 2 : 1) << ", true, " << kDirBlock << ", false, "
      << (geo.runs != 0 ? "true" : "false")
      << ">(D, " << (geo.runs != 0 ? "nullptr" : "cols.p") << ", " << (any_coded ? "dicts" : "nullptr") << ", n, "
      << (dev.filter_lds_off >= 0 ? "filter" : "nullptr") << ", view, DenseView{}, " << geo.dir_gids << ", 0, " << geo.nbuf << ", 1, "
      << (geo.runs != 0 ? "pieces" : "nullptr") << ", " << nulls_arg << ", &d);\n}\n}  // namespace qsx\n";
    return o.str();
  }
  // the run-of-blocks flavour of the body takes the same signature (the table arrives as `pieces`) and its stripes from the
  // table: its template arguments are emitted here, not patched into the text afterwards
  std::ostringstream body_args;
  body_args << "true, " << (dense ? "true" : "false") << ", " << num_sums << ", " << jit_rows_per_thread();
  if (geo.runs != 0 || geo.reg_groups != 0) {
    body_args << ", false, " << kABlock << ", false, " << (geo.runs != 0 ? "true" : "false");
    if (geo.reg_groups != 0) body_args << ", " << geo.reg_groups;   // per-wave register accumulators (agg_hash_update.hpp, REG)
  }
  o
    // explicit arguments are kept under 256 bytes (one view, the dictionaries behind a pointer): with the 256 hidden
    // bytes a kernarg segment beyond 512 bytes made the same code 2.4x slower (3.5 -> 8.3 ms, Q1 over 600 M rows)
    // One argument ORDER for every shape (cols, n, view, pieces, filter, dicts), but a shape only declares the trailing
    // ones it reads: the launcher always passes all six and hipModuleLaunchKernel copies what the kernel's metadata asks for.
    // Q1 without filter or dictionaries then has the kernarg segment of the AOT kernel (472 bytes; the geometry lives in the
    // shape as constants).  (The run-time Q1 shape still measures 1.26 against the AOT kernel's 1.16 ms per 200 M rows with
    // identical launch geometry and arguments: 5 % more instructions in the hipRTC build, cause not established.)
    // (geo.waves_per_eu: the occupancy the launcher wants from the registers — LDS would admit that many workgroups per CU.
    // Only asked of the compiler driver: hipRTC's pipeline spills for it, and a second in-process compile on a background
    // thread for the fallback crashed a short-lived host test)
    << "extern \"C\" __global__ __launch_bounds__(" << kABlock << (geo.waves_per_eu != 0 && !compiler_driver().empty() ? ", " + std::to_string(geo.waves_per_eu) : std::string())
    << ") void qsx_jit_agg(ColumnPointers cols, int64_t n,\n"
    << "    " << (dense ? "DenseView" : "HashTableView") << " view, const long long *pieces"
    << (dev.filter_lds_off >= 0 || any_coded || any_nulls ? ", const uint64_t *filter" : "") << (any_coded || any_nulls ? ", const void *const *dicts" : "")
    << nulls_param << ") {\n"
    << "  static constexpr DevConfig D = jit_make_dev();\n"
    << "  (void)cols;\n"
    << "  agg_hash_update_body<" << body_args.str() << ">(D, " << (geo.runs != 0 ? "nullptr" : "cols.p") << ", "
    << (any_coded ? "dicts" : "nullptr") << ", n, " << (dev.filter_lds_off >= 0 ? "filter" : "nullptr") << ", "
    << (dense ? "HashTableView{}, view" : "view, DenseView{}")
    << ", " << geo.S << ", " << geo.rep_shift << ", " << geo.nbuf << ", " << geo.ranges << ", pieces" << (any_nulls ? ", nulls" : "")
    << ");\n}\n}  // namespace qsx\n";
  return o.str();
}

bool jit_enabled() {
  static const bool on = []() {
    const char *e = getenv("QSX_AGG_JIT");
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

}  // namespace

// One per distinct source text.  state: 0 = compiling, 1 = ready, -1 = failed.
struct JitRequest {
  std::atomic<int> state{0};
  JitKernel *kernel = nullptr;
};

namespace {
// The cache, its lock and the compile threads are never destroyed: a compile may still be running when the process
// tears its statics down; an atexit hook waits for the threads instead.
std::mutex &cache_mutex() { static std::mutex *m = new std::mutex; return *m; }
// keyed by (device, source text): hipModuleLoadData loads the code object on the device that is current in the loading
// thread, and a hipFunction_t of one device's module must not be launched on another
std::map<std::pair<int, std::string>, JitRequest *> &cache() {
  static auto *c = new std::map<std::pair<int, std::string>, JitRequest *>;
  return *c;
}
std::vector<std::thread> &compile_threads() { static auto *t = new std::vector<std::thread>; return *t; }
void join_compile_threads() {
  std::vector<std::thread> threads;
  {
    std::lock_guard<std::mutex> lock(cache_mutex());
    threads.swap(compile_threads());
  }
  for (std::thread &t : threads) {
    if (t.joinable()) t.join();
  }
}

// ---- the code objects on disk (QSX_JIT_CACHE_DIR) ---------------------------------------------------------------------
// hipRTC takes 1-2 s per plan shape — longer than most queries run — so within one process a new shape is served by the
// interpreter kernel until its compile finishes.  A server that sees the same plan shapes after every restart sets
// QSX_JIT_CACHE_DIR: code objects are kept there, one file per (compiler version, source text), and a shape found on
// disk is loaded before the first update call returns.  A file holds the full source text in front of the code object
// and is only used when that text matches (the file name is a hash); files are written to a temporary name and renamed,
// so concurrent processes see whole files or none.  Unset (the default): no file is read or written.
constexpr char kCacheMagic[8] = {'Q', 'S', 'X', 'J', 'I', 'T', '0', '1'};
// The compiler driver of the ROCm install (hipcc), or empty: QSX_JIT_COMPILER = "hiprtc" (never spawn a compiler), a path, or
// unset = $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc, then hipcc on PATH — the first that is executable.
std::string compiler_driver() {
  const char *e = getenv("QSX_JIT_COMPILER");
  if (e != nullptr && std::strcmp(e, "hiprtc") == 0) return std::string();
  std::vector<std::string> candidates;
  if (e != nullptr && e[0] != '\0') {
    candidates.push_back(e);
  } else {
    if (const char *rocm = getenv("ROCM_PATH")) candidates.push_back(std::string(rocm) + "/bin/hipcc");
    candidates.push_back("/opt/rocm/bin/hipcc");
    if (const char *path = getenv("PATH")) {   // (a host whose Makefile found hipcc on PATH finds it here too)
      const std::string dirs(path);
      for (size_t at = 0; at <= dirs.size();) {
        const size_t end = dirs.find(':', at);
        const std::string dir = dirs.substr(at, end == std::string::npos ? std::string::npos : end - at);
        if (!dir.empty()) candidates.push_back(dir + "/hipcc");
        if (end == std::string::npos) break;
        at = end + 1;
      }
    }
  }
  for (const std::string &c : candidates) {
    if (access(c.c_str(), X_OK) == 0) return c;
  }
  return std::string();
}
std::string cache_stamp() {
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  const std::string driver = compiler_driver();
  // (the driver's KIND and the HIP version, not its path: code objects shipped with the library stay valid on a box whose
  // hipcc lives elsewhere)
  return (driver.empty() ? std::string("hiprtc ") : std::string("driver hipcc hip ")) + std::to_string(major) + "." + std::to_string(minor) +
         " gfx950 -O3 -ffp-contract=off -munsafe-fp-atomics\n";
}
std::string cache_file_name(const std::string &text) {
  unsigned long long h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;   // two FNV-1a walks with different offsets
  for (unsigned char c : text) {
    h1 = (h1 ^ c) * 0x100000001b3ull;
    h2 = (h2 ^ (c + 0x9Eu)) * 0x100000001b3ull;
  }
  char name[64];
  std::snprintf(name, sizeof(name), "/qsx_%016llx%016llx", h1, h2);
  return name;
}
// The directory code objects are WRITTEN to: QSX_JIT_CACHE_DIR, or none.
std::string cache_path(const std::string &stamped_source) {
  const char *dir = getenv("QSX_JIT_CACHE_DIR");
  if (dir == nullptr || dir[0] == '\0') return std::string();
  return std::string(dir) + cache_file_name(stamped_source) + ".hsaco";
}
// The code objects that ship with the library: <directory of libqsx.so>/jit_cache, filled by the build for the plan shapes
// listed under csrc/jit_shapes/ (__graft_entry__.build -> qsx_jit_warm).  Read-only, looked at after QSX_JIT_CACHE_DIR: an
// installation starts with the shapes its tests and benchmarks use already compiled, and nothing is ever written next to the
// library at run time.  QSX_JIT_SHIPPED_CACHE=0 ignores it (tests of the compile path).
std::string shipped_cache_dir() {
  const char *e = getenv("QSX_JIT_SHIPPED_CACHE");   // (read per call: tests switch it)
  if (e != nullptr && e[0] == '0') return std::string();
  static const std::string dir = []() {
    Dl_info info{};
    if (dladdr(reinterpret_cast<const void *>(&shipped_cache_dir), &info) == 0 || info.dli_fname == nullptr) return std::string();
    std::string path(info.dli_fname);
    const size_t slash = path.rfind('/');
    if (slash == std::string::npos) return std::string();
    return path.substr(0, slash) + "/jit_cache";
  }();
  return !dir.empty() && access(dir.c_str(), R_OK | X_OK) == 0 ? dir : std::string();
}
bool read_cache_file(const std::string &path, const std::string &stamped, std::string *code);
bool load_cached_code(const std::string &source, std::string *code) {
  const std::string stamped = cache_stamp() + source;
  const std::string own = cache_path(stamped);
  if (!own.empty() && read_cache_file(own, stamped, code)) return true;
  const std::string shipped = shipped_cache_dir();
  return !shipped.empty() && read_cache_file(shipped + cache_file_name(stamped) + ".hsaco", stamped, code);
}
bool read_cache_file(const std::string &path, const std::string &stamped, std::string *code) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (f == nullptr) return false;
  bool ok = false;
  char magic[8];
  unsigned long long source_len = 0, code_len = 0;
  if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kCacheMagic, 8) == 0 && std::fread(&source_len, 8, 1, f) == 1 &&
      std::fread(&code_len, 8, 1, f) == 1 && source_len == stamped.size() && code_len > 0 && code_len < (1ull << 30)) {
    std::string text(source_len, '\0');
    code->assign(code_len, '\0');
    ok = std::fread(&text[0], 1, source_len, f) == source_len && text == stamped &&
         std::fread(&(*code)[0], 1, code_len, f) == code_len;
  }
  std::fclose(f);
  return ok;
}
void write_cache_file(const std::string &path, const std::string &stamped, const std::string &code);
void store_cached_code(const std::string &source, const std::string &code) {
  const std::string stamped = cache_stamp() + source;
  const std::string path = cache_path(stamped);
  if (path.empty()) return;
  write_cache_file(path, stamped, code);
}
void write_cache_file(const std::string &path, const std::string &stamped, const std::string &code) {
  char suffix[48];
  std::snprintf(suffix, sizeof(suffix), ".tmp.%ld.%llx", static_cast<long>(getpid()),
                static_cast<unsigned long long>(std::hash<std::thread::id>()(std::this_thread::get_id())));
  const std::string tmp = path + suffix;
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (f == nullptr) return;                       // an unwritable directory only costs the next process a compile
  const unsigned long long source_len = stamped.size(), code_len = code.size();
  const bool ok = std::fwrite(kCacheMagic, 1, 8, f) == 8 && std::fwrite(&source_len, 8, 1, f) == 1 && std::fwrite(&code_len, 8, 1, f) == 1 &&
                  std::fwrite(stamped.data(), 1, stamped.size(), f) == stamped.size() &&
                  std::fwrite(code.data(), 1, code.size(), f) == code.size();
  if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

// Source text -> gfx950 code object: from QSX_JIT_CACHE_DIR when it is there, else hipRTC (and then into the directory).
// log (optional) receives the compiler's messages on failure.
// The source through the compiler driver, in a child process: hipRTC's pipeline builds measurably slower code from the same
// text (Q1's shape: no 128-bit LDS reads, the flush loop not unrolled, scalar registers spilled into vector lanes — 1.31
// against 1.18 ms per 200 M rows, tools/ab/jit_offline_ab.py; option for option the same as far as hiprtcCompileProgram
// takes them).  Any failure — no driver, no temporary directory, a non-zero exit — leaves the job to hipRTC.
bool compile_with_driver(const std::string &source, std::string *code, std::string *log) {
  const std::string driver = compiler_driver();
  if (driver.empty()) return false;
  const char *tmp = getenv("TMPDIR");
  std::string dir = std::string(tmp != nullptr && tmp[0] != '\0' ? tmp : "/tmp") + "/qsx_jit_XXXXXX";
  if (mkdtemp(&dir[0]) == nullptr) return false;
  const std::string src = dir + "/shape.hip", out = dir + "/shape.hsaco", messages = dir + "/messages.txt";
  bool ok = false;
  if (FILE *f = std::fopen(src.c_str(), "wb")) {
    ok = std::fwrite(source.data(), 1, source.size(), f) == source.size();
    ok = std::fclose(f) == 0 && ok;
  }
  if (ok) {
    // (the prelude's typedefs repeat <cstdint>'s: legal, same types; hip_runtime.h brings what hipRTC has built in)
    const char *argv[] = {driver.c_str(), "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "-munsafe-fp-atomics", "-include", "hip/hip_runtime.h", "-o", "shape.hsaco", "shape.hip", nullptr};
    posix_spawn_file_actions_t actions;
    posix_spawn_file_actions_init(&actions);
    // run inside the directory, on relative names: its random name stays out of the code object (same text, same bytes)
    posix_spawn_file_actions_addchdir_np(&actions, dir.c_str());
    posix_spawn_file_actions_addopen(&actions, 1, messages.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&actions, 1, 2);
    pid_t pid = 0;
    int status = -1;
    // The child gets the parent's environment MINUS everything that would load a tool library into it: under
    // `rocprofv3 -- python3 bench.py` the profiler's LD_PRELOAD / HSA_TOOLS_LIB / ROCP* variables would otherwise reach the
    // driver, whose preloaded library initialises the GPU there — and the driver then hops to clang and lld: a
    // GPU-initialised process replacing its program, which this pool treats as a hazard (and which, refused, silently turned
    // every profiled run into hipRTC-built code).  The compiler needs none of them.
    std::vector<char *> child_env;
    for (char **e = environ; e != nullptr && *e != nullptr; ++e) {
      static const char *const kDropped[] = {"LD_PRELOAD=", "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCP_", "ROCPROF", "ROCTRACER_",
                                             "ROCTX_", "HIP_TOOLS_", "OMPT_TOOL", "ROCM_TOOLS"};
      bool dropped = false;
      for (const char *prefix : kDropped) dropped = dropped || std::strncmp(*e, prefix, std::strlen(prefix)) == 0;
      if (!dropped) child_env.push_back(*e);
    }
    child_env.push_back(nullptr);
    ok = posix_spawn(&pid, driver.c_str(), &actions, nullptr, const_cast<char *const *>(argv), child_env.data()) == 0;
    posix_spawn_file_actions_destroy(&actions);
    if (ok) {
      pid_t waited = -1;
      while ((waited = waitpid(pid, &status, 0)) < 0 && errno == EINTR) {
      }
      // (a host that ignores SIGCHLD reaps the child itself: no status to read then — whether the code object is there decides)
      ok = waited < 0 ? errno == ECHILD : (WIFEXITED(status) && WEXITSTATUS(status) == 0);
    }
  }
  if (ok) {
    ok = false;
    if (FILE *f = std::fopen(out.c_str(), "rb")) {
      std::fseek(f, 0, SEEK_END);
      const long size = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      if (size > 0) {
        code->assign(static_cast<size_t>(size), '\0');
        ok = std::fread(&(*code)[0], 1, static_cast<size_t>(size), f) == static_cast<size_t>(size);
      }
      std::fclose(f);
    }
  } else if (log != nullptr) {
    if (FILE *f = std::fopen(messages.c_str(), "rb")) {
      char buf[4096];
      const size_t got = std::fread(buf, 1, sizeof(buf), f);
      log->assign(buf, got);
      std::fclose(f);
    }
  }
  (void)std::remove(src.c_str());
  (void)std::remove(out.c_str());
  (void)std::remove(messages.c_str());
  (void)rmdir(dir.c_str());
  return ok;
}

// QSX_JIT_RECORD_DIR: every plan shape this process asks for leaves its text there — the part behind the bundled kernel
// sources: the translated configuration and the entry point, a few KiB — as qsx_<hash>.shape.  That is how csrc/jit_shapes/
// is made (tools/record_jit_shapes.sh runs the GPU suite and the benchmarks with it): the build compiles what it lists.
size_t bundle_prefix_length() { return std::strlen(kPrelude) + std::strlen(kBundle); }
void record_shape(const std::string &source) {
  const char *dir = getenv("QSX_JIT_RECORD_DIR");
  if (dir == nullptr || dir[0] == '\0' || source.size() <= bundle_prefix_length()) return;
  const std::string tail = source.substr(bundle_prefix_length());
  const std::string path = std::string(dir) + cache_file_name(tail) + ".shape";
  if (access(path.c_str(), F_OK) == 0) return;
  char suffix[48];
  std::snprintf(suffix, sizeof(suffix), ".tmp.%ld.%llx", static_cast<long>(getpid()),
                static_cast<unsigned long long>(std::hash<std::thread::id>()(std::this_thread::get_id())));
  const std::string tmp = path + suffix;
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (f == nullptr) return;
  const bool ok = std::fwrite(tail.data(), 1, tail.size(), f) == tail.size();
  if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

bool compile_to_code(const std::string &source, std::string *code, std::string *log) {
  record_shape(source);
  if (load_cached_code(source, code)) return true;
  if (compile_with_driver(source, code, log)) {
    store_cached_code(source, *code);
    return true;
  }
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, source.c_str(), "qsx_jit_agg.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
  std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
                                    "-mllvm", "-amdgpu-internalize-symbols"};   // the last two: what hipcc passes for device code
  // QSX_JIT_OPTIONS: further options, blank-separated (experiments with what hipRTC's pipeline does differently from hipcc's)
  std::vector<std::string> extra;
  if (const char *e = getenv("QSX_JIT_OPTIONS")) {
    std::istringstream words(e);
    for (std::string w; words >> w;) extra.push_back(w);
    for (const std::string &w : extra) opts.push_back(w.c_str());
  }
  const hiprtcResult rc = hiprtcCompileProgram(prog, static_cast<int>(opts.size()), opts.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string text(log_size, '\0');
    if (log_size != 0) hiprtcGetProgramLog(prog, &text[0]);
    if (log != nullptr) *log = std::string(hiprtcGetErrorString(rc)) + "\n" + text;
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t code_size = 0;
  hiprtcGetCodeSize(prog, &code_size);
  code->assign(code_size, '\0');
  hiprtcGetCode(prog, &(*code)[0]);
  hiprtcDestroyProgram(&prog);
  store_cached_code(source, *code);
  return true;
}

JitKernel *load_code(const std::string &code);

JitKernel *compile(const std::string &source) {
  std::string code, log;
  if (!compile_to_code(source, &code, &log)) {
    std::fprintf(stderr, "[qsx] run-time plan shape: hipRTC failed; the interpreter kernel is used.\n%.2000s\n", log.c_str());
    return nullptr;
  }
  return load_code(code);
}

JitKernel *load_code(const std::string &code) {
  JitKernel *k = new JitKernel();
  if (hipModuleLoadData(&k->module, code.data()) != hipSuccess ||
      hipModuleGetFunction(&k->function, k->module, "qsx_jit_agg") != hipSuccess) {
    (void)hipGetLastError();
    std::fprintf(stderr, "[qsx] run-time plan shape: loading the code object failed; the interpreter kernel is used.\n");
    delete k;
    return nullptr;
  }
  // up to the CU's 160 KiB of dynamic LDS (same opt-in as the AOT kernels; harmless if the runtime ignores it)
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k->function), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipGetLastError();
  return k;
}

// Plan shapes run without scratch memory (their state lives in registers and LDS: DESIGN.md §4).  A shape built for a
// requested occupancy (JitGeometry::waves_per_eu) that had to spill for it is rebuilt from `relaxed` (the same shape without
// the request: scratch traffic costs more than the workgroup it buys); a shape that needs scratch even then is not used at all
// — the interpreter kernel serves its state (the one such build of round 4, a register struct kept in memory by a pointer
// comparison, also produced wrong sums through hipRTC).
int scratch_bytes_of(const JitKernel *k) {
  int scratch = 0;
  if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->function) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return scratch;
}
JitKernel *settle(JitKernel *k, const std::string &relaxed) {
  if (k == nullptr || scratch_bytes_of(k) == 0) return k;
  // (nothing was ever launched from the rejected module — it was loaded to read its attributes only — so it is unloaded
  // here, by the thread that loaded it)
  if (k->module != nullptr) (void)hipModuleUnload(k->module);
  delete k;
  if (relaxed.empty()) {
    std::fprintf(stderr, "[qsx] run-time plan shape: the build needs scratch memory; the interpreter kernel is used.\n");
    return nullptr;
  }
  std::string cached;
  return settle(load_cached_code(relaxed, &cached) ? load_code(cached) : compile(relaxed), std::string());
}

}  // namespace

// Source text only (tests / offline inspection; needs no device).
std::string jit_agg_source(const DevConfig &dev, int num_sums, bool dense, const JitGeometry &geometry) {
  return make_source(dev, num_sums, dense, geometry);
}

JitRequest *jit_agg_request(const DevConfig &dev, int num_sums, bool dense, const JitGeometry &geometry, bool synchronous) {
  if (!jit_enabled()) return nullptr;
  // Without a compiler driver the shape is built by hipRTC IN this process — then in the calling thread: a process that exits
  // while a background thread is inside hiprtcCompileProgram tears LLVM's statics down under it (they are registered at
  // hipRTC's first use, after this file's exit hook, and so destroyed before it runs: a short-lived host test crashed that
  // way).  The driver's compile is a child process; its thread only waits.
  synchronous = synchronous || compiler_driver().empty();
  const std::string source = make_source(dev, num_sums, dense, geometry);   // a filter is part of dev (filter_lds_off)
  std::string relaxed;
  if (geometry.waves_per_eu != 0 && !compiler_driver().empty()) {
    JitGeometry without = geometry;
    without.waves_per_eu = 0;
    relaxed = make_source(dev, num_sums, dense, without);
  }
  JitRequest *r = nullptr;
  bool mine = false;
  int device = 0;
  (void)hipGetDevice(&device);
  {
    std::lock_guard<std::mutex> lock(cache_mutex());
    const std::pair<int, std::string> key(device, source);
    auto it = cache().find(key);
    if (it != cache().end()) {
      r = it->second;
    } else {
      r = new JitRequest();
      cache().emplace(key, r);
      mine = true;
      std::string cached;
      if (!synchronous) record_shape(source);
      if (!synchronous && load_cached_code(source, &cached)) {
        // on disk (QSX_JIT_CACHE_DIR): a file read and a module load, milliseconds — ready before the first launch
        JitKernel *k = settle(load_code(cached), relaxed);
        r->kernel = k;
        r->state.store(k != nullptr ? 1 : -1, std::memory_order_release);
        mine = false;
      } else if (!synchronous) {
        // hipRTC takes 1-2 s: the caller keeps using the interpreter kernel and picks the shape up when it is ready
        static bool hooked = false;
        if (!hooked) {
          std::atexit(join_compile_threads);
          hooked = true;
        }
        compile_threads().emplace_back([r, source, relaxed, device]() {
          (void)hipSetDevice(device);
          JitKernel *k = settle(compile(source), relaxed);
          r->kernel = k;
          r->state.store(k != nullptr ? 1 : -1, std::memory_order_release);
        });
      }
    }
  }
  if (mine && synchronous) {
    JitKernel *k = settle(compile(source), relaxed);
    r->kernel = k;
    r->state.store(k != nullptr ? 1 : -1, std::memory_order_release);
  } else if (synchronous) {
    while (r->state.load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));   // someone else compiles it
  }
  return r;
}

int jit_request_state(JitRequest *r, const JitKernel **kernel) {
  const int state = r->state.load(std::memory_order_acquire);
  if (kernel != nullptr) *kernel = state == 1 ? r->kernel : nullptr;
  return state;
}

int jit_agg_launch(const JitKernel *k, int grid, size_t lds_bytes, hipStream_t stream, const ColumnPointers &cols,
                   const void *const *dict_table_dev, int64_t n, const uint64_t *filter, const HashTableView &g,
                   const DenseView &dense, bool is_dense, int S, int rep_shift, int nbuf, int ranges, const long long *pieces, int block,
                   const unsigned long long *const *null_table_dev) {
  ColumnPointers a_cols = cols;
  const unsigned long long *const *a_nulls = null_table_dev;
  const void *const *a_dicts = dict_table_dev;
  int64_t a_n = n;
  const uint64_t *a_filter = filter;
  HashTableView a_g = g;
  DenseView a_dense = dense;
  const long long *a_pieces = pieces;
  void *view = is_dense ? static_cast<void *>(&a_dense) : static_cast<void *>(&a_g);
  (void)S; (void)rep_shift; (void)nbuf; (void)ranges;   // (constants inside the shape)
  void *args[] = {&a_cols, &a_n, view, &a_pieces, &a_filter, &a_dicts, &a_nulls};   // (a shape declares a prefix of these: make_source)
  QSX_HIP_TRY(hipModuleLaunchKernel(k->function, static_cast<unsigned>(grid), 1, 1, static_cast<unsigned>(block), 1, 1,
                                    static_cast<unsigned>(lds_bytes), stream, args, nullptr));
  return QSX_OK;
}

bool jit_compiles_out_of_process() { return !compiler_driver().empty(); }

int jit_rows_per_thread() {
  static const int rows = []() {
    const char *e = getenv("QSX_AGG_JIT_ROWS");
    return e != nullptr && atoi(e) == 2 ? 2 : 4;
  }();
  return rows;
}

int jit_resident_blocks(const JitKernel *k, int block, size_t lds_bytes) {
  int blocks = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k->function, block, lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return blocks;
}

int jit_agg_launch_dir(const JitKernel *k, int grid, size_t lds_bytes, hipStream_t stream, const ColumnPointers &cols,
                       const void *const *dict_table_dev, int64_t n, const uint64_t *filter, const HashTableView &g, const DirView &d,
                       const long long *pieces, const unsigned long long *const *null_table_dev) {
  ColumnPointers a_cols = cols;
  const unsigned long long *const *a_nulls = null_table_dev;
  const void *const *a_dicts = dict_table_dev;
  int64_t a_n = n;
  const uint64_t *a_filter = filter;
  HashTableView a_g = g;
  DirView a_d = d;
  const long long *a_pieces = pieces;
  void *args[] = {&a_cols, &a_dicts, &a_n, &a_filter, &a_g, &a_d, &a_pieces, &a_nulls};
  QSX_HIP_TRY(hipModuleLaunchKernel(k->function, static_cast<unsigned>(grid), 1, 1, kDirBlock, 1, 1,
                                    static_cast<unsigned>(lds_bytes), stream, args, nullptr));
  return QSX_OK;
}

}  // namespace qsx

// Build step (not part of include/qsx.h; called by __graft_entry__.build for every file of csrc/jit_shapes/): compile the plan
// shape whose recorded text is `tail` with the compiler driver and keep the code object in out_dir under the name the
// run-time lookup computes (shipped_cache_dir).  1: already there, 0: compiled, < 0: a status.  Needs no GPU.
extern "C" int qsx_jit_warm(const char *tail, size_t tail_bytes, const char *out_dir) {
  using namespace qsx;
  if (tail == nullptr || out_dir == nullptr || tail_bytes == 0) return QSX_ERR_INVALID_ARGUMENT;
  const std::string source = std::string(kPrelude) + kBundle + std::string(tail, tail_bytes);
  const std::string stamped = cache_stamp() + source;
  const std::string path = std::string(out_dir) + cache_file_name(stamped) + ".hsaco";
  std::string code, log;
  if (read_cache_file(path, stamped, &code)) return 1;
  if (compiler_driver().empty()) return QSX_ERR_UNSUPPORTED;     // (a hipRTC build would carry another stamp: nothing to ship)
  if (!compile_with_driver(source, &code, &log)) {
    std::fprintf(stderr, "qsx_jit_warm: %.2000s\n", log.c_str());
    return QSX_ERR_HIP;
  }
  write_cache_file(path, stamped, code);
  return 0;
}

// Test hook (not part of include/qsx.h): compiles the plan shape of a configuration with hipRTC and
// reports whether that worked — runs without a GPU, so the CPU test suite covers the generator.
extern "C" int qsx_debug_jit_compile(const qsx_agg_config_t *config, int with_filter, size_t *out_code_bytes) {
  using namespace qsx;
  if (config == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  Translated t = translate(*config);
  if (t.status != QSX_OK) return t.status;
  const bool directory = (with_filter & 2) != 0 && !t.dense;   // bit 1: the group-directory variant
  const bool runs = (with_filter & 4) != 0;                    // bit 2: the run-of-blocks flavour
  const bool dense_lds = (with_filter & 8) != 0 && t.dense;    // bit 3: a dense state in LDS
  const bool reg_groups = (with_filter & 16) != 0;             // bit 4: per-wave register accumulators (small hash tables)
  with_filter &= 1;
  plan_tile(t.dev, t.used_columns, directory || dense_lds ? kDirBlock : kABlock * jit_rows_per_thread(), with_filter != 0,
            /*reg_decode=*/!directory && !dense_lds);
  // a plausible geometry: this hook only checks that the shape compiles
  JitGeometry geometry = directory ? JitGeometry{4096, 0, 2, 1, 4096, 0} : JitGeometry{t.dense ? 8 : 16, t.dense ? 0 : 4, 1, 1, 0, 0};
  if (dense_lds) geometry = JitGeometry{4096, 0, 2, 2, 4096, 0};
  geometry.runs = runs ? 1 : 0;
  if (reg_groups && !directory && !t.dense) geometry.reg_groups = reg_groups_for(geometry.S, t.num_sums);
  const std::string source = jit_agg_source(t.dev, t.num_sums, t.dense, geometry);
  if (const char *dump = getenv("QSX_JIT_DUMP")) {   // the generated translation unit, for offline inspection with hipcc -S
    if (FILE *f = std::fopen(dump, "w")) {
      std::fwrite(source.data(), 1, source.size(), f);
      std::fclose(f);
    }
  }
  std::string code, log;
  const bool compiled = compile_to_code(source, &code, &log);
  const size_t code_size = code.size();
  if (compiled) {
    if (const char *dump = getenv("QSX_JIT_DUMP_CODE")) {   // the code object, for llvm-objdump / llvm-readelf --notes
      if (FILE *f = std::fopen(dump, "wb")) {
        std::fwrite(code.data(), 1, code.size(), f);
        std::fclose(f);
      }
    }
  } else {
    std::fprintf(stderr, "%.4000s\n", log.c_str());
  }
  if (out_code_bytes != nullptr) *out_code_bytes = code_size;
  return compiled ? QSX_OK : QSX_ERR_HIP;
}
