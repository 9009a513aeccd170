// extern "C" entry points of libttemb_hip.so (declared in include/ttemb.h).
// Argument checking, workspace carving and kernel-family dispatch live here; the
// kernels are in ttemb_generic.hip / ttemb_fast3.hip / ttemb_cache.hip.
#include "ttemb_common.h"
#include "ttemb_cache.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ttemb {

static thread_local char g_err[512] = "";
// process-wide on purpose: autograd runs backward on its own thread
static std::atomic<int> g_path{TTEMB_PATH_AUTO};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return TTEMB_OK;
  return fail(TTEMB_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

int allow_big_lds(const void* kernel, size_t lds_bytes, LdsGate* gate, const char* what) {
  if (lds_bytes <= kLdsDefault) return TTEMB_OK;
  if (lds_bytes > kCuLds)
    return fail(TTEMB_E_UNSUPPORTED, "%s needs %lld bytes of LDS per workgroup, the CU has %lld", what, (long long)lds_bytes, (long long)kCuLds);
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rc) return rc;
  const uint64_t bit = dev >= 0 && dev < 64 ? (uint64_t(1) << dev) : 0;
  if (bit && (gate->devices.load(std::memory_order_acquire) & bit)) return TTEMB_OK;
  rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCuLds), what);
  if (rc == TTEMB_OK && bit) gate->devices.fetch_or(bit, std::memory_order_release);
  return rc;
}

int device_cus() {
  static std::atomic<int> cus[64];   // zero-initialised: "not asked yet"
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  const bool slot = dev >= 0 && dev < 64;
  if (slot && (n = cus[dev].load(std::memory_order_relaxed)) > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (slot) cus[dev].store(n, std::memory_order_relaxed);
  return n;
}

int current_path() { return g_path.load(); }

// ---- device-side faults (a bounded wait of the grouping pass that ran out: report_fault in ttemb_fast3.hip) ----
// One pinned, device-visible host word per process, allocated by ttemb_init() (or the first ttemb_status()) -- an explicit
// call the host makes outside any stream capture, never by a lookup (the lookups allocate nothing) -- and never freed (its
// address is baked into captured graphs).  The kernel that gives up stores its reason there; every lookup entry point looks
// at it first (one atomic exchange on a host word, no synchronisation), so the fault surfaces as TTEMB_E_HIP on the next
// call the host makes after the store has landed -- at the latest on the one after a synchronisation -- and is consumed by
// exactly one caller.  The error therefore names an EARLIER call: the call that reports it has not been started.  The
// results of the faulted call itself are NaN (poisoned plan), whether or not anyone asks.
static std::atomic<uint32_t*> g_fault_host{nullptr};
static std::atomic<uint32_t*> g_fault_dev{nullptr};
static std::atomic<int> g_fault_state{0};   // 0 = not tried, 1 = being set up, 2 = ready, 3 = not available

// the lookups' view: the word's device address when it exists, else null (that call reports through its NaN results only)
uint32_t* fault_word(hipStream_t) {
  return g_fault_state.load(std::memory_order_acquire) == 2 ? g_fault_dev.load(std::memory_order_relaxed) : nullptr;
}

// ttemb_init / ttemb_status: create the word (once per process; a call that loses the race goes without)
int fault_word_init() {
  int state = g_fault_state.load(std::memory_order_acquire);
  if (state == 2) return TTEMB_OK;
  if (state == 3) return fail(TTEMB_E_HIP, "no pinned host memory for the device-fault word: expired device-side waits are reported through NaN results only");
  int expect = 0;
  if (!g_fault_state.compare_exchange_strong(expect, 1)) return TTEMB_OK;   // another thread is setting it up
  void* host = nullptr;
  void* dev = nullptr;
  if (hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || host == nullptr ||
      hipHostGetDevicePointer(&dev, host, 0) != hipSuccess || dev == nullptr) {
    (void)hipGetLastError();
    g_fault_state.store(3, std::memory_order_release);
    return fail(TTEMB_E_HIP, "no pinned host memory for the device-fault word: expired device-side waits are reported through NaN results only");
  }
  memset(host, 0, 64);
  g_fault_host.store(reinterpret_cast<uint32_t*>(host), std::memory_order_relaxed);
  g_fault_dev.store(reinterpret_cast<uint32_t*>(dev), std::memory_order_relaxed);
  g_fault_state.store(2, std::memory_order_release);
  return TTEMB_OK;
}

int pending_device_fault() {
  if (g_fault_state.load(std::memory_order_acquire) != 2) return TTEMB_OK;
  uint32_t* w = g_fault_host.load(std::memory_order_relaxed);
  if (__atomic_load_n(w, __ATOMIC_RELAXED) == 0u) return TTEMB_OK;   // (the usual case: one plain read)
  const uint32_t code = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);   // consumed by exactly one caller; a report landing now stays for the next
  if (code == 0u) return TTEMB_OK;
  return fail(TTEMB_E_HIP,
              "an EARLIER grouped lookup of this process gave up waiting on the device (%s): the rows / gradients of that call are "
              "NaN, not wrong numbers; a fused SGD / Adagrad backward on such a plan left the parameters and the optimizer state "
              "untouched (dense gradients are NaN).  The GPU is shared or throttled beyond what the grouping pass tolerates.  The "
              "call that returns this error was not started; discard the faulted step's outputs and run it again",
              code == 1u ? "range counter take-over in the decode step" : "look-back of the place step");
}

static std::atomic<bool> g_prof_on{false};
// slots: 0 forward chain kernel | 1 all backward chain kernels | 2 backward chunk kernel | 3 grouping pass (+ prefix products) |
//        4 cache probe pass (lookup, + LFU update when fused) | 5 partition scatter | 6 cached-row gather | 7 cached-row update |
//        8 group epilogue kernel | 9 finalize kernel
constexpr int kProfSlots = 10;
static hipEvent_t g_prof_ev[kProfSlots][2] = {};
static std::atomic<bool> g_prof_valid[kProfSlots] = {};

void profile_begin(int which, hipStream_t st) {
  if (!g_prof_on) return;
  for (int i = 0; i < 2; ++i)
    if (g_prof_ev[which][i] == nullptr && hipEventCreate(&g_prof_ev[which][i]) != hipSuccess) return;
  g_prof_valid[which] = hipEventRecord(g_prof_ev[which][0], st) == hipSuccess;
}

void profile_end(int which, hipStream_t st) {
  if (!g_prof_on || !g_prof_valid[which]) return;
  g_prof_valid[which] = hipEventRecord(g_prof_ev[which][1], st) == hipSuccess;
}

int make_dev_shape(const ttemb_shape_t* s, DevShape* out) {
  if (s == nullptr) return fail(TTEMB_E_BADARG, "shape is null");
  if (s->T < 2 || s->T > TTEMB_MAX_CORES)
    return fail(TTEMB_E_BADARG, "T=%d: the layer supports 2..4 cores", s->T);
  DevShape d;
  memset(&d, 0, sizeof(d));
  d.T = s->T;
  if (s->R[0] != 1 || s->R[s->T] != 1) return fail(TTEMB_E_BADARG, "R[0] and R[T] must be 1");
  long long D = 1;
  for (int t = 0; t < s->T; ++t) {
    if (s->p[t] <= 0 || s->q[t] <= 0 || s->R[t] <= 0 || s->R[t + 1] <= 0)
      return fail(TTEMB_E_BADARG, "non-positive factor at core %d", t);
    d.p[t] = s->p[t];
    d.q[t] = s->q[t];
    d.R[t] = s->R[t];
    D *= s->q[t];
    if (D > (1 << 20)) return fail(TTEMB_E_BADARG, "embedding_dim too large");
  }
  d.R[s->T] = 1;
  if (D % 4 != 0) return fail(TTEMB_E_BADARG, "embedding_dim %lld must be a multiple of 4", D);
  d.D = (int)D;
  long long L = 1;
  for (int t = s->T - 1; t >= 0; --t) {
    d.L[t] = L;
    if (L > (1ll << 62) / d.p[t]) return fail(TTEMB_E_BADARG, "prod(p) overflows");
    L *= d.p[t];
  }
  long long Q = 1;
  int pm = 0;
  for (int t = 0; t < s->T; ++t) {
    Q *= d.q[t];
    long long rl = (long long)d.R[t] * d.q[t] * d.R[t + 1];
    long long pl = Q * d.R[t + 1];
    if (rl > (1 << 24) || pl > (1 << 24)) return fail(TTEMB_E_BADARG, "core row too large");
    d.row_len[t] = (int)rl;
    d.part_len[t] = (int)pl;
    if (t + 1 < s->T && d.part_len[t] > pm) pm = d.part_len[t];
  }
  d.part_max = pm;
  *out = d;
  return TTEMB_OK;
}

static int64_t grad_scratch_bytes(const DevShape& s) {
  int64_t b = 0;
  for (int t = 0; t < s.T; ++t) b += align256((int64_t)s.p[t] * s.row_len[t] * 4);
  return b;
}

// have_offsets: the call carries its bag boundaries (size queries assume so) -- what a call past one 32-bit row window needs
// to run on the grouped path, piece by piece
static bool use_fast3(const DevShape& s, int64_t nnz, int64_t B, bool have_offsets = true) {
  const int path = current_path();
  if (path == TTEMB_PATH_GENERIC || path == TTEMB_PATH_PER_BAG || !fast3_supported(s)) return false;
  if (!fast3_fits(s, nnz, B) && !(have_offsets && fast3_fits_in_pieces(s, nnz, B))) return false;
  return path == TTEMB_PATH_FAST3 || fast3_pays(s, nnz);
}
// the fused optimiser step rides in the grouped backward's last kernel when the call is one piece
static bool one_piece(const DevShape& s, int64_t nnz, int64_t B) { return fast3_fits(s, nnz, B); }

// small batches of an instantiated 3-core shape whose ids come with their bag boundaries: one wavefront per bag, MFMA per
// id (ttemb_small3.inc) instead of the wave-per-id scalar kernels
static bool use_small3(const DevShape& s, int64_t nnz, int64_t B, const int64_t* rowidx, const int64_t* offsets) {
  const int path = current_path();
  return (path == TTEMB_PATH_AUTO || path == TTEMB_PATH_PER_BAG) && rowidx == nullptr && offsets != nullptr && small3_supported(s) &&
         !use_fast3(s, nnz, B, offsets != nullptr);
}

// ---------------------------------------------------------------------------------
// 4-core tables on the grouped path.  row = G0[i0].G1[i1].G2[i2].G3[i3] is a 3-core row over the table
// (p0 p1, p2, p3) with the VIRTUAL first core V[(i0, i1)] = G0[i0].G1[i1]  (q0 q1 x r2) -- or over (p0, p1, p2 p3)
// with the virtual last core V[(i2, i3)] = G2[i2].G3[i3]: the id digits are the same (i0 p1 + i1 is the leading,
// i2 p3 + i3 the trailing digit of the 3-core split), so the grouped kernels run unchanged on (V, G2, G3) /
// (G0, G1, V) when the merged (q, ranks) is one of their shapes.  V is rebuilt from the cores per call (a few
// thousand small GEMMs), its gradient is split back:  dA[ia] = sum_ib dV[ia,ib].B[ib]^T, dB[ib] = sum_ia A[ia]^T.dV[ia,ib]
// -- two tiny kernels around the 3-core path.
// ---------------------------------------------------------------------------------
struct Merged4 {
  bool on;
  bool per_bag;         // the 3-core view runs on the per-bag kernels (no grouped shape fits / the batch is small)
  int a;                // the merged pair: cores a and a + 1 (0: the first two, 2: the last two)
  DevShape s3;          // the 3-core view
  int64_t v_bytes;      // bytes of V (and of dV), 256-aligned
  // V[(ia, ib)] = A[ia] (rows x K) . Bm[ib] (K x n): the two cores as plain matrices
  int pa, pb, rows, K, n;
};

static bool view3(const DevShape& s, int a, DevShape* out) {   // the 3-core shape with cores a, a + 1 merged
  const long long pp = (long long)s.p[a] * s.p[a + 1], qq = (long long)s.q[a] * s.q[a + 1];
  if (pp > 0x7fffffffll || qq > 1024) return false;
  DevShape d;
  memset(&d, 0, sizeof(d));
  d.T = 3;
  int k = 0;
  for (int t = 0; t < 4; ++t) {
    if (t == a + 1) continue;
    d.p[k] = t == a ? (int)pp : s.p[t];
    d.q[k] = t == a ? (int)qq : s.q[t];
    d.R[k] = s.R[t];
    ++k;
  }
  d.R[3] = 1;
  d.D = s.D;
  d.L[2] = 1; d.L[1] = d.p[2]; d.L[0] = (long long)d.p[1] * d.p[2];
  long long Q = 1;
  int pm = 0;
  for (int t = 0; t < 3; ++t) {
    Q *= d.q[t];
    const long long rl = (long long)d.R[t] * d.q[t] * d.R[t + 1], pl = Q * d.R[t + 1];
    if (rl > (1 << 24) || pl > (1 << 24)) return false;
    d.row_len[t] = (int)rl;
    d.part_len[t] = (int)pl;
    if (t < 2 && d.part_len[t] > pm) pm = d.part_len[t];
  }
  d.part_max = pm;
  *out = d;
  return true;
}

// first choice: merge the first two cores (small virtual core, the per-id operand stays the last core); else the
// last two (q = 5,5,2,2: the virtual last core has p2 p3 rows of r2 q2 q3 floats, its dG slabs grow with it)
// 2-core tables ride on the same 3-core kernels: row = G0[i0].G1[i1] = G0[i0].I.G1[i1] is a 3-core row over
// (p0, 1, p1) with q = (q0, 1, q1), ranks (r1, r1) and the r1 x r1 identity as the only row of a VIRTUAL middle core
// (a == -1 below).  The groups are the values of i0, the "prefix product" of a group is its G0 row, the gradient of
// the identity is computed and dropped.  (FBTT/tt_embeddings_cuda.cu:757-779, :81-117 are the reference's 2-core forms.)
static bool view_lifted2(const DevShape& s, DevShape* out) {
  DevShape d;
  memset(&d, 0, sizeof(d));
  d.T = 3;
  d.p[0] = s.p[0]; d.p[1] = 1; d.p[2] = s.p[1];
  d.q[0] = s.q[0]; d.q[1] = 1; d.q[2] = s.q[1];
  d.R[0] = 1; d.R[1] = s.R[1]; d.R[2] = s.R[1]; d.R[3] = 1;
  d.D = s.D;
  d.L[2] = 1; d.L[1] = d.p[2]; d.L[0] = (long long)d.p[1] * d.p[2];
  long long Q = 1;
  int pm = 0;
  for (int t = 0; t < 3; ++t) {
    Q *= d.q[t];
    const long long rl = (long long)d.R[t] * d.q[t] * d.R[t + 1], pl = Q * d.R[t + 1];
    if (rl > (1 << 24) || pl > (1 << 24)) return false;
    d.row_len[t] = (int)rl;
    d.part_len[t] = (int)pl;
    if (t < 2 && d.part_len[t] > pm) pm = d.part_len[t];
  }
  d.part_max = pm;
  *out = d;
  return true;
}

// per_bag_ok: the call could run on the per-bag kernels (ids + offsets, no row index; sizing queries pass true): a 2- or
// 4-core table whose 3-core view has no grouped shape, or whose batch is below the grouped path's crossover, then rides on
// the per-bag MFMA kernels (templated or run-time shape) through the same virtual core instead of the scalar kernels.
static Merged4 merge_first_two(const DevShape& s, int64_t nnz, int64_t B, bool per_bag_ok = false, bool have_offsets = true) {
  Merged4 m;
  memset(&m, 0, sizeof(m));
  const int path = current_path();
  if (path == TTEMB_PATH_GENERIC) return m;
  per_bag_ok = per_bag_ok && (path == TTEMB_PATH_AUTO || path == TTEMB_PATH_PER_BAG);
  if (s.T == 2) {
    DevShape d;
    if (!view_lifted2(s, &d)) return m;
    const bool f3 = use_fast3(d, nnz, B, have_offsets);
    if (f3 || (per_bag_ok && small3_supported(d))) {
      m.on = true;
      m.per_bag = !f3;
      m.a = -1;
      m.s3 = d;
      m.K = s.R[1];
      m.v_bytes = align256((long long)m.K * m.K * 4);
    }
    return m;
  }
  if (s.T != 4) return m;
  auto fill = [&](int a, const DevShape& d, bool per_bag) {
    m.on = true;
    m.per_bag = per_bag;
    m.a = a;
    m.s3 = d;
    m.pa = s.p[a]; m.pb = s.p[a + 1];
    m.rows = s.R[a] * s.q[a];          // A[ia] is (R_a q_a) x R_{a+1}
    m.K = s.R[a + 1];
    m.n = s.q[a + 1] * s.R[a + 2];     // Bm[ib] is R_{a+1} x (q_{a+1} R_{a+2})
    m.v_bytes = align256((long long)m.pa * m.pb * m.rows * m.n * 4);
  };
  for (int a = 0; a <= 2; a += 2) {
    DevShape d;
    if (!view3(s, a, &d) || !use_fast3(d, nnz, B, have_offsets)) continue;
    if (a == 2 && (long long)d.p[2] * d.row_len[2] * 4 > (4ll << 20)) continue;   // virtual last core: at most 4 MB (its slabs)
    fill(a, d, false);
    return m;
  }
  if (per_bag_ok) {   // first pair merged: the per-id operand stays the last core
    // The virtual core is REBUILT by every forward and split back (over all p0 p1 rows, gradients zero-filled) by every
    // backward, whatever nnz is: O(p0 p1 row) next to the scalar kernels' O(nnz).  The run scripts' 4-core tables have
    // 3 000-row virtual cores (0.4-1.5 MB: a few microseconds); a table with a large first pair takes this view only when
    // the batch amortises the rebuild -- at least one id per 16 rows of V -- or V is small anyway (<= 4 MB).  A conservative
    // gate, not a measured crossover (no script trains such a table).
    DevShape d;
    if (view3(s, 0, &d) && small3_supported(d)) {
      const long long v_rows = (long long)s.p[0] * s.p[1], v_bytes = v_rows * d.row_len[0] * 4;
      if (v_bytes <= (256ll << 20) && (v_bytes <= (4ll << 20) || nnz * 16 >= v_rows)) fill(0, d, true);
    }
  }
  return m;
}

// ---------------------------------------------------------------------------------
// Ranks off the instantiated list on the grouped path: zero-padded cores.
// A 3-core table with ranks (r1, r2) IS the table with ranks (R, R), R >= r1, r2, whose cores carry zeros in the added rank
// positions: every row is the same number for number (the added terms are products with zero), and the gradient with respect
// to an original entry is the same sum.  So any rank in [2, 256] -- tuning_SAGE.py:213 searches exactly that interval --
// rides on the grouped MFMA kernels of the next instantiated rank of its q shape (8 / 16 / 32; 64 / 128 / 256 for the wide
// chain): per call the cores are copied into padded form (a few hundred KB to a few MB: one small launch), the grouped
// kernels run on the padded table, and the backward keeps the original sub-block of the padded gradient.  The extra flops
// ((R / r)^2 on the first contraction) are small change next to what the per-bag kernels pay at large batches (no prefix
// reuse, float atomics per id): rank 12 at 65 536 ids -- 0.48 ms per-bag -- runs like rank 16.
// ---------------------------------------------------------------------------------
struct Padded3 {
  bool on;
  DevShape sp;               // the padded 3-core shape
  int64_t core_bytes[3];     // bytes of every padded core, 256-aligned
  int64_t cores_total;       // their sum
};

static Padded3 pad_ranks(const DevShape& s, int64_t nnz, int64_t B, bool have_offsets) {
  Padded3 pd;
  memset(&pd, 0, sizeof(pd));
  const int path = current_path();
  if (s.T != 3 || path == TTEMB_PATH_GENERIC || path == TTEMB_PATH_PER_BAG || fast3_supported(s)) return pd;
  const int need = s.R[1] > s.R[2] ? s.R[1] : s.R[2];
  for (int R : {8, 16, 32, 64, 128, 256}) {
    if (R < need) continue;
    ttemb_shape_t t;
    memset(&t, 0, sizeof(t));
    t.T = 3;
    for (int k = 0; k < 3; ++k) { t.p[k] = s.p[k]; t.q[k] = s.q[k]; }
    t.R[0] = 1; t.R[1] = R; t.R[2] = R; t.R[3] = 1;
    DevShape d;
    if (make_dev_shape(&t, &d) != TTEMB_OK || !fast3_supported(d)) continue;
    if (!use_fast3(d, nnz, B, have_offsets)) return pd;   // (a larger rank of the list would pay even later)
    pd.on = true;
    pd.sp = d;
    for (int k = 0; k < 3; ++k) {
      pd.core_bytes[k] = align256((int64_t)d.p[k] * d.row_len[k] * 4);
      pd.cores_total += pd.core_bytes[k];
    }
    return pd;
  }
  return pd;
}

// dst[row][a][j][b] (ranks Ra x Rb) = src[row][a][j][b] inside the original ranks (ra x rb), 0 outside -- and back
struct PadJob {
  const float* src[3];
  float* dst[3];
  int p[3], q[3], ra[3], rb[3], Ra[3], Rb[3];
};
__global__ __launch_bounds__(256) void pad_cores_kernel(PadJob j) {
  const int t = blockIdx.y;
  const long long n = (long long)j.p[t] * j.Ra[t] * j.q[t] * j.Rb[t];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int b = (int)(e % j.Rb[t]);
    long long r = e / j.Rb[t];
    const int jj = (int)(r % j.q[t]);
    r /= j.q[t];
    const int a = (int)(r % j.Ra[t]);
    const long long row = r / j.Ra[t];
    j.dst[t][e] = (a < j.ra[t] && b < j.rb[t]) ? j.src[t][((row * j.ra[t] + a) * j.q[t] + jj) * j.rb[t] + b] : 0.f;
  }
}
// the original sub-block of a padded gradient: src is padded (Ra x Rb), dst original (ra x rb)
__global__ __launch_bounds__(256) void unpad_cores_kernel(PadJob j) {
  const int t = blockIdx.y;
  const long long n = (long long)j.p[t] * j.ra[t] * j.q[t] * j.rb[t];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int b = (int)(e % j.rb[t]);
    long long r = e / j.rb[t];
    const int jj = (int)(r % j.q[t]);
    r /= j.q[t];
    const int a = (int)(r % j.ra[t]);
    const long long row = r / j.ra[t];
    j.dst[t][e] = j.src[t][((row * j.Ra[t] + a) * j.q[t] + jj) * j.Rb[t] + b];
  }
}
static PadJob pad_job(const DevShape& s, const Padded3& pd) {
  PadJob j;
  memset(&j, 0, sizeof(j));
  for (int t = 0; t < 3; ++t) {
    j.p[t] = s.p[t]; j.q[t] = s.q[t];
    j.ra[t] = s.R[t]; j.rb[t] = s.R[t + 1];
    j.Ra[t] = pd.sp.R[t]; j.Rb[t] = pd.sp.R[t + 1];
  }
  return j;
}
// padded copies of the cores at `buf` (pd.cores_total bytes); *cp3 receives their pointers
static int build_padded_cores(const DevShape& s, const Padded3& pd, const CorePtrs& cp, char* buf, CorePtrs* cp3, hipStream_t st) {
  PadJob j = pad_job(s, pd);
  memset(cp3, 0, sizeof(*cp3));
  int64_t off = 0, most = 0;
  for (int t = 0; t < 3; ++t) {
    j.src[t] = cp.c[t];
    j.dst[t] = reinterpret_cast<float*>(buf + off);
    cp3->c[t] = j.dst[t];
    off += pd.core_bytes[t];
    const int64_t n = (int64_t)pd.sp.p[t] * pd.sp.row_len[t];
    most = n > most ? n : most;
  }
  int64_t blocks = (most + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(pad_cores_kernel, dim3((unsigned)blocks, 3), dim3(256), 0, st, j);
  return check_hip(hipGetLastError(), "pad_cores_kernel");
}
static int unpad_grads(const DevShape& s, const Padded3& pd, const CorePtrsMut& padded, const CorePtrsMut& dst, hipStream_t st) {
  PadJob j = pad_job(s, pd);
  int64_t most = 0;
  for (int t = 0; t < 3; ++t) {
    j.src[t] = padded.c[t];
    j.dst[t] = dst.c[t];
    const int64_t n = (int64_t)s.p[t] * s.row_len[t];
    most = n > most ? n : most;
  }
  int64_t blocks = (most + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(unpad_cores_kernel, dim3((unsigned)blocks, 3), dim3(256), 0, st, j);
  return check_hip(hipGetLastError(), "unpad_cores_kernel");
}

// V[(ia, ib)] = A[ia] (rows x K) . Bm[ib] (K x n): one workgroup per pair
__global__ __launch_bounds__(256) void merge_pair_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int pb, int rows,
                                                         int K, int n, float* __restrict__ V) {
  const int pair = blockIdx.x, ia = pair / pb, ib = pair - ia * pb;
  const float* a = A + (size_t)ia * rows * K;
  const float* b = Bm + (size_t)ib * K * n;
  float* v = V + (size_t)pair * rows * n;
  for (int e = threadIdx.x; e < rows * n; e += 256) {
    const int r = e / n, c = e - r * n;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(a[r * K + k], b[k * n + c], acc);
    v[e] = acc;
  }
}

// `outs` (< 128) outputs of `terms` products each: 256 / (outs rounded up to a power of two) threads share an output
template <typename F>
__device__ __forceinline__ void sum_terms_few(int outs, int terms, float* dst, float* red, F term) {
  const int tid = threadIdx.x;
  int lanes = 256;
  for (int o2 = 1; o2 < outs; o2 <<= 1) lanes >>= 1;
  const int o = tid / lanes, l = tid - o * lanes;
  float acc = 0.f;
  if (o < outs)
    for (int k = l; k < terms; k += lanes) acc += term(o, k);
  red[tid] = acc;
  __syncthreads();
  for (int w = lanes >> 1; w > 0; w >>= 1) {
    if (l < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (l == 0 && o < outs) dst[o] = red[tid];
}

// grid (pa + pb, K): workgroup (ia, k) sums dA[ia][:, k] = sum over ib of dV[ia,ib] . Bm[ib][k, :]^T, workgroup (pa + ib, k)
// sums dBm[ib][k, :] = sum over ia of A[ia][:, k]^T . dV[ia,ib]   (one column / row of K per workgroup: the sums are short
// but there are only pa + pb of each kind, so they are spread over K times as many workgroups)
__global__ __launch_bounds__(256) void split_pair_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const float* __restrict__ dV, int pa, int pb, int rows, int K, int n,
                                                         float* __restrict__ dA, float* __restrict__ dBm) {
  __shared__ float red[256];
  __shared__ float outv[256];
  const int k = blockIdx.y;
  if ((int)blockIdx.x < pa) {
    const int ia = blockIdx.x;
    for (int r0 = 0; r0 < rows; r0 += 127) {   // outputs r, at most 127 per pass
      const int nr = rows - r0 < 127 ? rows - r0 : 127;
      sum_terms_few(nr, pb * n, outv, red, [&](int o, int t) {
        const int r = r0 + o, ib = t / n, c = t - ib * n;
        return dV[((size_t)ia * pb + ib) * rows * n + r * n + c] * Bm[((size_t)ib * K + k) * n + c];
      });
      __syncthreads();
      for (int o = threadIdx.x; o < nr; o += 256) dA[((size_t)ia * rows + r0 + o) * K + k] = outv[o];
      __syncthreads();
    }
  } else {
    const int ib = blockIdx.x - pa;
    for (int c0 = 0; c0 < n; c0 += 127) {
      const int nc = n - c0 < 127 ? n - c0 : 127;
      sum_terms_few(nc, pa * rows, outv, red, [&](int o, int t) {
        const int c = c0 + o, ia = t / rows, r = t - ia * rows;
        return A[((size_t)ia * rows + r) * K + k] * dV[((size_t)ia * pb + ib) * rows * n + r * n + c];
      });
      __syncthreads();
      for (int o = threadIdx.x; o < nc; o += 256) dBm[((size_t)ib * K + k) * n + c0 + o] = outv[o];
      __syncthreads();
    }
  }
}

__global__ void identity_core_kernel(float* __restrict__ V, int K) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < K * K) V[e] = (e / K == e % K) ? 1.f : 0.f;
}

static int build_merged_core(const Merged4& m, const CorePtrs& cp, float* V, hipStream_t st) {
  if (m.a < 0) {   // the lifted 2-core table: the virtual middle core is the identity
    hipLaunchKernelGGL(identity_core_kernel, dim3((unsigned)((m.K * m.K + 255) / 256)), dim3(256), 0, st, V, m.K);
    return check_hip(hipGetLastError(), "identity_core_kernel");
  }
  hipLaunchKernelGGL(merge_pair_kernel, dim3((unsigned)(m.pa * m.pb)), dim3(256), 0, st, cp.c[m.a], cp.c[m.a + 1], m.pb, m.rows,
                     m.K, m.n, V);
  return check_hip(hipGetLastError(), "merge_pair_kernel");
}

// the 3-core operand lists of a merged table: V in the place of the pair
static void merged_cores(const Merged4& m, const CorePtrs& cp, const float* V, CorePtrs* c3) {
  memset(c3, 0, sizeof(*c3));
  if (m.a < 0) { c3->c[0] = cp.c[0]; c3->c[1] = V; c3->c[2] = cp.c[1]; }
  else if (m.a == 0) { c3->c[0] = V; c3->c[1] = cp.c[2]; c3->c[2] = cp.c[3]; }
  else          { c3->c[0] = cp.c[0]; c3->c[1] = cp.c[1]; c3->c[2] = V; }
}

__global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(p + i) & 15) == 0) {
      *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (size_t j = i; j < n && j < i + 4; ++j) p[j] = 0u;
    }
  }
}

struct ZeroSegs {
  uint32_t* p[TTEMB_MAX_CORES];
  size_t n[TTEMB_MAX_CORES];   // words
};

__global__ void zero_segments_kernel(ZeroSegs z, int T) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (int t = 0; t < T; ++t)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.n[t]; i += stride) z.p[t][i] = 0u;
}

int launch_zero_cores(const DevShape& s, const CorePtrsMut& d_cores, hipStream_t st) {
  ZeroSegs z;
  size_t most = 0;
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) {
    z.p[t] = t < s.T ? reinterpret_cast<uint32_t*>(d_cores.c[t]) : nullptr;
    z.n[t] = t < s.T ? (size_t)s.p[t] * s.row_len[t] : 0;
    most = z.n[t] > most ? z.n[t] : most;
  }
  if (most == 0) return TTEMB_OK;
  const size_t blocks = (most + 255) / 256;
  hipLaunchKernelGGL(zero_segments_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, z, s.T);
  return check_hip(hipGetLastError(), "zero d_cores");
}

int launch_zero(void* p, size_t bytes, hipStream_t st, const char* what) {
  if (bytes == 0) return TTEMB_OK;
  const size_t n = bytes / 4;
  size_t blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n);
  return check_hip(hipGetLastError(), what);
}

// rows whose bag length is not 1 must be zero before the lookups accumulate into them
__global__ void zero_rows_kernel(const int64_t* __restrict__ offsets, int64_t B, int D,
                                 float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (offsets[b + 1] - offsets[b] == 1) return;
  float4* o = reinterpret_cast<float4*>(out + b * D);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c * 4 < D; ++c) o[c] = z;
}

__global__ void sgd_step_kernel(float* __restrict__ w, const float* __restrict__ g, int64_t n, float lr, const float* __restrict__ skip) {
  // (ttemb_sgd_step_guarded: some rank's gradient came from a poisoned plan.  The word is tested BIT-wise: an all-reduced float
  //  count (k.0f) and the uint32 poison word of a workspace header (1) both read non-zero, +0.0f and 0u both zero)
  if (skip != nullptr && *reinterpret_cast<const uint32_t*>(skip) != 0u) return;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 wv = *reinterpret_cast<float4*>(w + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    wv.x -= lr * gv.x; wv.y -= lr * gv.y; wv.z -= lr * gv.z; wv.w -= lr * gv.w;
    *reinterpret_cast<float4*>(w + i) = wv;
  } else {
    for (; i < n; ++i) w[i] -= lr * g[i];
  }
}

__global__ void adagrad_step_kernel(float* __restrict__ w, float* __restrict__ st,
                                    const float* __restrict__ g, int64_t n, float lr, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gv = g[i];
  const float s2 = st[i] + gv * gv;
  st[i] = s2;
  w[i] -= lr * gv / (sqrtf(s2) + eps);
}

struct Seg3 {
  float* w[TTEMB_MAX_CORES];
  float* st[TTEMB_MAX_CORES];
  const float* g[TTEMB_MAX_CORES];
  long long n[TTEMB_MAX_CORES];
};

// one launch for every core: blockIdx.y selects the core
// `skip`: the poison word a grouped backward of this call left in the workspace header (FusedUpdate::poison_out), or null
__global__ void fused_step_kernel(Seg3 seg, float lr, float eps, int adagrad, const uint32_t* __restrict__ skip) {
  if (skip != nullptr && *skip != 0u) return;   // the gradients are NaN and the host hears of it: parameters stay as they are
  const int t = blockIdx.y;
  float* __restrict__ w = seg.w[t];
  const float* __restrict__ g = seg.g[t];
  const long long n = seg.n[t];
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      float4 wv = *reinterpret_cast<float4*>(w + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      if (adagrad) {
        float4 sv = *reinterpret_cast<float4*>(seg.st[t] + i);
        sv.x += gv.x * gv.x; sv.y += gv.y * gv.y; sv.z += gv.z * gv.z; sv.w += gv.w * gv.w;
        *reinterpret_cast<float4*>(seg.st[t] + i) = sv;
        wv.x -= lr * gv.x / (sqrtf(sv.x) + eps); wv.y -= lr * gv.y / (sqrtf(sv.y) + eps);
        wv.z -= lr * gv.z / (sqrtf(sv.z) + eps); wv.w -= lr * gv.w / (sqrtf(sv.w) + eps);
      } else {
        wv.x -= lr * gv.x; wv.y -= lr * gv.y; wv.z -= lr * gv.z; wv.w -= lr * gv.w;
      }
      *reinterpret_cast<float4*>(w + i) = wv;
    } else {
      for (long long j = i; j < n; ++j) {
        if (adagrad) {
          const float s2 = seg.st[t][j] + g[j] * g[j];
          seg.st[t][j] = s2;
          w[j] -= lr * g[j] / (sqrtf(s2) + eps);
        } else {
          w[j] -= lr * g[j];
        }
      }
    }
  }
}

static int run_sgd(float* w, const float* g, int64_t n, float lr, hipStream_t st, const float* skip = nullptr) {
  if (n <= 0) return TTEMB_OK;
  if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g)) & 15)
    return fail(TTEMB_E_BADARG, "sgd_step: buffers must be 16-byte aligned");
  const int threads = 256;
  const int64_t blocks = ((n + 3) / 4 + threads - 1) / threads;
  hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, w, g, n, lr, skip);
  return check_hip(hipGetLastError(), "sgd_step_kernel");
}

static int run_adagrad(float* w, float* state, const float* g, int64_t n, float lr, float eps, hipStream_t st) {
  if (n <= 0) return TTEMB_OK;
  const int threads = 256;
  const int64_t blocks = (n + threads - 1) / threads;
  hipLaunchKernelGGL(adagrad_step_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, w, state, g, n, lr, eps);
  return check_hip(hipGetLastError(), "adagrad_step_kernel");
}

static int check_lookup_args(const void* cores, const void* indices, int64_t nnz, int64_t B) {
  if (nnz < 0 || B < 0) return fail(TTEMB_E_BADARG, "negative size (nnz=%lld, B=%lld)", (long long)nnz, (long long)B);
  if (nnz > 0x7fffffffll) return fail(TTEMB_E_BADARG, "nnz=%lld exceeds int32 range", (long long)nnz);
  if (cores == nullptr) return fail(TTEMB_E_BADARG, "cores is null");
  if (nnz > 0 && indices == nullptr) return fail(TTEMB_E_BADARG, "indices is null");
  return TTEMB_OK;
}

// rows of the ids: the caller's rowidx, or (rowidx == NULL) derived from offsets into the head
// of the workspace; *ws / *ws_bytes are advanced past the part used
static int resolve_rowidx(const int64_t** rowidx, const int64_t* offsets, int64_t nnz, int64_t B, char** ws,
                          int64_t* ws_bytes, hipStream_t st, bool rows_in_plan = false) {
  const int64_t need = align256(nnz * 8);
  char* base = *ws;
  if (base != nullptr && *ws_bytes >= need) {
    *ws = base + need;
    *ws_bytes -= need;
  } else if (*rowidx == nullptr && nnz > 0) {
    return fail(TTEMB_E_WORKSPACE, "workspace too small for the row index (%lld bytes)", (long long)need);
  }
  if (*rowidx != nullptr || nnz == 0) return TTEMB_OK;
  if (offsets == nullptr) return fail(TTEMB_E_BADARG, "rowidx and offsets are both null");
  return TTEMB_OK;   // both kernel families derive the rows from `offsets` themselves (no expansion launch, no array)
}

// shared body of the three backward entry points: gradient of the live ids into `dst`
static int backward_into(const DevShape& ds, const CorePtrs& cp, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev, int64_t B,
                         const float* d_output, const CorePtrsMut& dst, void* ws, int64_t ws_bytes,
                         const void* plan, int64_t plan_bytes, hipStream_t st, void* header, const FusedUpdate* update = nullptr,
                         bool* grouped = nullptr) {
  // *grouped: the gradients come from a grouped backward, whose finalize kernel left its verdict in the header's poison word
  if (grouped != nullptr) *grouped = false;
  if (use_fast3(ds, nnz, B, offsets != nullptr)) {
    if (grouped != nullptr) *grouped = true;
    return launch_backward_fast3(ds, cp, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, dst, ws, ws_bytes, plan,
                                 plan_bytes, st, update, header);
  }
  const Merged4 m4 = merge_first_two(ds, nnz, B, rowidx == nullptr && offsets != nullptr, offsets != nullptr);
  if (m4.on) {   // 4 cores: the 3-core backward on (V, G2, G3), then dV back onto G0 and G1
    if (grouped != nullptr) *grouped = !m4.per_bag;
    if (ws == nullptr || ws_bytes < 2 * m4.v_bytes) return fail(TTEMB_E_WORKSPACE, "backward needs room for the merged core");
    float* V = reinterpret_cast<float*>(ws);
    float* dV = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + m4.v_bytes);
    int rc = build_merged_core(m4, cp, V, st);
    if (rc) return rc;
    CorePtrs c3;
    CorePtrsMut d3;
    merged_cores(m4, cp, V, &c3);
    memset(&d3, 0, sizeof(d3));
    if (m4.a < 0) { d3.c[0] = dst.c[0]; d3.c[1] = dV; d3.c[2] = dst.c[1]; }   // (the identity's gradient is dropped)
    else if (m4.a == 0) { d3.c[0] = dV; d3.c[1] = dst.c[2]; d3.c[2] = dst.c[3]; }
    else           { d3.c[0] = dst.c[0]; d3.c[1] = dst.c[1]; d3.c[2] = dV; }
    if (m4.per_bag) {   // the per-bag kernels ADD into the gradients: clear them (the real cores, then dV)
      rc = launch_zero_cores(ds, dst, st);
      if (rc == TTEMB_OK) rc = launch_zero(dV, (size_t)m4.v_bytes, st, "zero dV");
      if (rc == TTEMB_OK) rc = launch_backward_small3(m4.s3, c3, indices, offsets, nnz, nnz_dev, B, d_output, d3, st);
    } else {
      rc = launch_backward_fast3(m4.s3, c3, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, d3,
                                 reinterpret_cast<char*>(ws) + 2 * m4.v_bytes, ws_bytes - 2 * m4.v_bytes, plan, plan_bytes, st, nullptr, header);
    }
    if (rc || m4.a < 0) return rc;
    hipLaunchKernelGGL(split_pair_kernel, dim3((unsigned)(m4.pa + m4.pb), (unsigned)m4.K), dim3(256), 0, st, cp.c[m4.a], cp.c[m4.a + 1], dV, m4.pa,
                       m4.pb, m4.rows, m4.K, m4.n, dst.c[m4.a], dst.c[m4.a + 1]);
    return check_hip(hipGetLastError(), "split_pair_kernel");
  }
  const Padded3 pd = pad_ranks(ds, nnz, B, offsets != nullptr);
  if (pd.on) {   // ranks off the list: grouped backward on the padded cores, then the original sub-block of its gradient
    if (update != nullptr) return fail(TTEMB_E_BADARG, "internal: a padded table writes gradients, the step follows");
    if (grouped != nullptr) *grouped = true;
    if (ws == nullptr || ws_bytes < 2 * pd.cores_total) return fail(TTEMB_E_WORKSPACE, "backward needs room for the padded cores");
    char* wp = reinterpret_cast<char*>(ws);
    CorePtrs cp3;
    int rc = build_padded_cores(ds, pd, cp, wp, &cp3, st);
    if (rc) return rc;
    CorePtrsMut d3;
    memset(&d3, 0, sizeof(d3));
    int64_t off = pd.cores_total;
    for (int t = 0; t < 3; ++t) {
      d3.c[t] = reinterpret_cast<float*>(wp + off);
      off += pd.core_bytes[t];
    }
    rc = launch_backward_fast3(pd.sp, cp3, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, d3, wp + 2 * pd.cores_total,
                               ws_bytes - 2 * pd.cores_total, plan, plan_bytes, st, nullptr, header);
    if (rc) return rc;
    return unpad_grads(ds, pd, d3, dst, st);
  }
  if (current_path() == TTEMB_PATH_FAST3) return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  int rc = launch_zero_cores(ds, dst, st);
  if (rc) return rc;
  if (use_small3(ds, nnz, B, rowidx, offsets))
    return launch_backward_small3(ds, cp, indices, offsets, nnz, nnz_dev, B, d_output, dst, st);
  return launch_backward_generic(ds, cp, indices, rowidx, offsets, B, nnz, nnz_dev, d_output, dst, st);
}

}  // namespace ttemb

using namespace ttemb;

// roctx ranges around the entry points (what the reference's drivers get from torch.profiler around the extension calls,
// sage_profiler.py): with TTEMB_ROCTX=1 in the environment every lookup / cache entry point is bracketed by
// roctxRangePush / roctxRangePop, so a rocprofv3 --marker-trace timeline shows the calls above their kernels.  The marker
// library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so): no link dependency, and nothing
// but one relaxed load per call when the variable is not set.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("TTEMB_ROCTX");
    if (e == nullptr || e[0] == '0' || e[0] == '\0') return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push != nullptr && pop != nullptr) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
const Roctx& roctx() {
  static const Roctx r;
  return r;
}
struct ApiRange {
  bool on;
  explicit ApiRange(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~ApiRange() {
    if (on) roctx().pop();
  }
  ApiRange(const ApiRange&) = delete;
  ApiRange& operator=(const ApiRange&) = delete;
};
}  // namespace

extern "C" {

int ttemb_abi_version(void) { return TTEMB_ABI_VERSION; }

const char* ttemb_last_error(void) { return g_err; }

int ttemb_set_path(int32_t path) {
  if (path < TTEMB_PATH_AUTO || path > TTEMB_PATH_PER_BAG) return fail(TTEMB_E_BADARG, "unknown path %d", path);
  g_path.store(path);
  return TTEMB_OK;
}

int ttemb_set_piece_limits(int64_t rows, int64_t ids) {
  fast3_set_piece_limits(rows, ids);
  return TTEMB_OK;
}

int ttemb_set_wide_slab_min_ids(int64_t ids) {
  fast3_set_wide_slab_min_ids(ids);
  return TTEMB_OK;
}

int ttemb_set_spin_limit(int64_t tries) {
  fast3_set_spin_limit(tries);
  return TTEMB_OK;
}

int ttemb_init(void) { return fault_word_init(); }

int ttemb_status(void) {
  if (g_fault_state.load(std::memory_order_acquire) == 0) (void)fault_word_init();   // (a caller that asks wants the word to exist)
  return pending_device_fault();
}

int ttemb_profile_enable(int32_t on) {
  if (on != 0)   // a read must never return a bracket recorded before this enable (by another leg, another kernel family)
    for (int i = 0; i < kProfSlots; ++i) g_prof_valid[i] = false;
  g_prof_on.store(on != 0);
  return TTEMB_OK;
}

int ttemb_profile_read(int32_t which, float* ms_host) {
  if (which < 0 || which >= kProfSlots || ms_host == nullptr) return fail(TTEMB_E_BADARG, "bad profile slot");
  if (!g_prof_valid[which]) return fail(TTEMB_E_BADARG, "no profiled launch recorded for slot %d", which);
  int rc = check_hip(hipEventSynchronize(g_prof_ev[which][1]), "hipEventSynchronize");
  if (rc) return rc;
  return check_hip(hipEventElapsedTime(ms_host, g_prof_ev[which][0], g_prof_ev[which][1]), "hipEventElapsedTime");
}

int64_t ttemb_workspace_bytes(const ttemb_shape_t* shape, int32_t op, int64_t nnz, int64_t B) {
  if (nnz < 0 || B < 0) return fail(TTEMB_E_BADARG, "negative size");
  // EVERY op leaves the first kFast3HeaderBytes of its workspace alone: the grouped lookup keeps its few persistent words
  // there, and callers reuse one workspace for all ops
  if (op == TTEMB_OP_PREPROCESS) return kFast3HeaderBytes + preprocess_workspace_bytes(nnz);
  DevShape ds;
  int rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  const bool f3 = use_fast3(ds, op == TTEMB_OP_CACHE_POPULATE ? B : nnz, B);
  const Merged4 m4 = op == TTEMB_OP_CACHE_POPULATE ? Merged4{} : merge_first_two(ds, nnz, B, true);
  // every lookup op's workspace begins with the header the grouped path keeps its few persistent words in
  const Padded3 pd = (op == TTEMB_OP_CACHE_POPULATE || f3 || m4.on) ? Padded3{} : pad_ranks(ds, nnz, B, true);
  if (pd.on && op == TTEMB_OP_FORWARD)
    return kFast3HeaderBytes + align256(nnz * 8) + pd.cores_total + fast3_workspace_bytes(pd.sp, op, nnz, B);
  if (pd.on && op == TTEMB_OP_BACKWARD)
    return kFast3HeaderBytes + grad_scratch_bytes(ds) + align256(nnz * 8) + 2 * pd.cores_total + fast3_workspace_bytes(pd.sp, op, nnz, B);
  switch (op) {
    case TTEMB_OP_FORWARD:
      if (m4.on) return kFast3HeaderBytes + align256(nnz * 8) + m4.v_bytes + (m4.per_bag ? 0 : fast3_workspace_bytes(m4.s3, op, nnz, B));
      return kFast3HeaderBytes + align256(nnz * 8) + (f3 ? fast3_workspace_bytes(ds, op, nnz, B) : 0);
    case TTEMB_OP_BACKWARD:
      if (m4.on) return kFast3HeaderBytes + grad_scratch_bytes(ds) + align256(nnz * 8) + 2 * m4.v_bytes + (m4.per_bag ? 0 : fast3_workspace_bytes(m4.s3, op, nnz, B));
      return kFast3HeaderBytes + grad_scratch_bytes(ds) + align256(nnz * 8) + (f3 ? fast3_workspace_bytes(ds, op, nnz, B) : 0);
    case TTEMB_OP_CACHE_POPULATE: {
      const int64_t sort = populate_workspace_bytes(nnz);
      if (sort < 0) return fail(TTEMB_E_HIP, "rocprim size query failed");
      return kFast3HeaderBytes + sort + (f3 ? fast3_workspace_bytes(ds, TTEMB_OP_FORWARD, B, B) : 0);
    }
    default:
      return fail(TTEMB_E_BADARG, "unknown op %d", op);
  }
}

int ttemb_kernel_family(const ttemb_shape_t* shape, int64_t nnz, int64_t B, int32_t ids_with_offsets) {
  DevShape ds;
  int rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  if (nnz < 0 || B < 0) return fail(TTEMB_E_BADARG, "negative size");
  auto family3 = [nnz, B](const DevShape& s, bool grouped) {
    if (grouped) return fast3_wide(s) ? (int)TTEMB_FAMILY_GROUPED_WIDE : (TTEMB_FAMILY_GROUPED | (fast3_prefix_in_chain(s, nnz, B) ? TTEMB_FAMILY_PREFIX_IN_CHAIN : 0) |
                                                                          (fast3_group_products_in_chain(s, nnz, B) ? TTEMB_FAMILY_GROUP_PRODUCTS_IN_CHAIN : 0));
    return small3_templated_shape(s) ? (int)TTEMB_FAMILY_PER_BAG : (int)TTEMB_FAMILY_PER_BAG_RT;
  };
  // (without the bag boundaries a call past one row window cannot be cut into pieces)
  if (use_fast3(ds, nnz, B, ids_with_offsets != 0)) return family3(ds, true);
  const Merged4 m4 = merge_first_two(ds, nnz, B, ids_with_offsets != 0, ids_with_offsets != 0);
  if (m4.on) return family3(m4.s3, !m4.per_bag) | TTEMB_FAMILY_MERGED;
  const Padded3 pd = pad_ranks(ds, nnz, B, ids_with_offsets != 0);
  if (pd.on) return family3(pd.sp, true) | TTEMB_FAMILY_PADDED;
  // use_small3 with stand-in pointers: only their null-ness is looked at
  const int64_t* none = nullptr;
  const int64_t* some = reinterpret_cast<const int64_t*>(&ds);
  if (use_small3(ds, nnz, B, ids_with_offsets ? none : some, ids_with_offsets ? some : none)) return family3(ds, false);
  return TTEMB_FAMILY_SCALAR;
}

int64_t ttemb_plan_bytes(const ttemb_shape_t* shape, int64_t nnz) {
  DevShape ds;
  int rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  if (nnz < 0) return fail(TTEMB_E_BADARG, "negative size");
  // (a call that runs in pieces keeps no plan -- a plan describes one piece --; neither does a per-bag view)
  if (use_fast3(ds, nnz, 0)) return fast3_fits(ds, nnz, 0) ? fast3_plan_bytes(ds, nnz) : 0;
  const Merged4 m4 = merge_first_two(ds, nnz, 0);
  if (m4.on) return fast3_fits(m4.s3, nnz, 0) ? fast3_plan_bytes(m4.s3, nnz) : 0;
  const Padded3 pd = pad_ranks(ds, nnz, 0, true);   // ranks off the list: the plan of the padded table
  return pd.on && fast3_fits(pd.sp, nnz, 0) ? fast3_plan_bytes(pd.sp, nnz) : 0;
}

}  // extern "C"

// phase 0 = the whole forward; 1 = everything that depends only on the ids; 2 = the rest (reads the cores)
static int forward_phase(int phase, const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                         const int32_t* nnz_dev, int64_t B, float* output, void* workspace,
                         int64_t workspace_bytes, void* plan, int64_t plan_bytes, void* stream) {
  DevShape ds;
  int rc = pending_device_fault();
  if (rc) return rc;
  rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  rc = check_lookup_args(cores, indices, nnz, B);
  if (rc) return rc;
  if (B == 0) return TTEMB_OK;
  if (output == nullptr) return fail(TTEMB_E_BADARG, "output is null");
  if (B >= 0x7fffffffll) return fail(TTEMB_E_BADARG, "B exceeds int32 range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  CorePtrs cp;
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) cp.c[t] = t < ds.T ? cores[t] : nullptr;
  void* header = nullptr;   // the grouped path's persistent words: the first kFast3HeaderBytes of every lookup workspace
  if (workspace != nullptr && workspace_bytes >= kFast3HeaderBytes) {
    header = workspace;
    workspace = reinterpret_cast<char*>(workspace) + kFast3HeaderBytes;
    workspace_bytes -= kFast3HeaderBytes;
  }
  const Merged4 m4 = nnz > 0 ? merge_first_two(ds, nnz, B, rowidx == nullptr && offsets != nullptr, offsets != nullptr) : Merged4{};
  if (m4.on && m4.per_bag && phase == 1) return TTEMB_OK;   // the per-bag kernels have no id-only half
  if (m4.on) {   // 4 cores through the 3-core kernels on (V = G0.G1, G2, G3); workspace: [row slot | V | 3-core]
    char* w4 = reinterpret_cast<char*>(workspace);
    const int64_t head = align256(nnz * 8);
    if (w4 == nullptr || workspace_bytes < head + m4.v_bytes) return fail(TTEMB_E_WORKSPACE, "forward needs room for the merged core");
    if (rowidx == nullptr && offsets == nullptr) return fail(TTEMB_E_BADARG, "rowidx and offsets are both null");
    float* V = reinterpret_cast<float*>(w4 + head);
    CorePtrs c3;
    merged_cores(m4, cp, V, &c3);
    if (phase != 1) {
      rc = build_merged_core(m4, cp, V, st);
      if (rc) return rc;
    }
    if (m4.per_bag)   // one launch, every output row written once
      return launch_forward_small3(m4.s3, c3, indices, offsets, nnz, nnz_dev, B, output, st);
    if (phase != 2 && offsets == nullptr) {
      rc = launch_zero(output, (size_t)B * ds.D * 4, st, "zero output");
      if (rc) return rc;
    }
    return launch_forward_fast3(m4.s3, c3, indices, rowidx, offsets, nnz, nnz_dev, B, output, offsets != nullptr,
                                w4 + head + m4.v_bytes, workspace_bytes - head - m4.v_bytes, plan, plan_bytes, phase, st, header);
  }
  const bool f3 = nnz > 0 && use_fast3(ds, nnz, B, offsets != nullptr);
  const Padded3 pd = (nnz > 0 && !f3) ? pad_ranks(ds, nnz, B, offsets != nullptr) : Padded3{};
  if (pd.on) {   // ranks off the instantiated list: the grouped kernels on zero-padded cores; workspace: [row slot | padded cores | grouped]
    char* wp = reinterpret_cast<char*>(workspace);
    const int64_t head = align256(nnz * 8);
    if (wp == nullptr || workspace_bytes < head + pd.cores_total) return fail(TTEMB_E_WORKSPACE, "forward needs room for the padded cores");
    if (rowidx == nullptr && offsets == nullptr) return fail(TTEMB_E_BADARG, "rowidx and offsets are both null");
    CorePtrs cp3;
    memset(&cp3, 0, sizeof(cp3));
    if (phase != 1) {   // (the id-only half does not read the cores)
      rc = build_padded_cores(ds, pd, cp, wp + head, &cp3, st);
      if (rc) return rc;
    }
    if (phase != 2 && offsets == nullptr) {
      rc = launch_zero(output, (size_t)B * ds.D * 4, st, "zero output");
      if (rc) return rc;
    }
    return launch_forward_fast3(pd.sp, cp3, indices, rowidx, offsets, nnz, nnz_dev, B, output, offsets != nullptr,
                                wp + head + pd.cores_total, workspace_bytes - head - pd.cores_total, plan, plan_bytes, phase, st, header);
  }
  if (phase == 1 && !f3) return TTEMB_OK;   // the generic kernels have no id-only half: phase 2 is their whole forward
  if (phase == 2 && f3) {
    return launch_forward_fast3(ds, cp, indices, rowidx, offsets, nnz, nnz_dev, B, output, offsets != nullptr, workspace,
                                workspace_bytes, plan, plan_bytes, 2, reinterpret_cast<hipStream_t>(stream), header);
  }
  if (rowidx == nullptr && offsets == nullptr && nnz > 0) return fail(TTEMB_E_BADARG, "rowidx and offsets are both null");
  if (!f3 && use_small3(ds, nnz, B, rowidx, offsets))   // one launch: every output row written once, zeros for an empty bag
    return launch_forward_small3(ds, cp, indices, offsets, nnz, nnz_dev, B, output, st);
  char* ws = reinterpret_cast<char*>(workspace);
  // the row-index slot at the head of the workspace is part of the layout on both paths; the fast path derives
  // rows (and clears the rows of bags that do not hold exactly one id) inside its grouping pass
  rc = resolve_rowidx(&rowidx, offsets, nnz, B, &ws, &workspace_bytes, st, f3);
  if (rc) return rc;
  workspace = ws;
  if (offsets == nullptr) {
    rc = launch_zero(output, (size_t)B * ds.D * 4, st, "zero output");
  } else if (!f3) {
    const int threads = 256;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((B + threads - 1) / threads)), dim3(threads), 0,
                       st, offsets, B, ds.D, output);
    rc = check_hip(hipGetLastError(), "zero_rows_kernel");
  }
  if (rc || nnz == 0) return rc;
  if (f3)
    return launch_forward_fast3(ds, cp, indices, rowidx, offsets, nnz, nnz_dev, B, output, offsets != nullptr, workspace,
                                workspace_bytes, plan, plan_bytes, phase, st, header);
  if (current_path() == TTEMB_PATH_FAST3) return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  return launch_forward_generic(ds, cp, indices, rowidx, offsets, B, nnz, nnz_dev, output, st);
}

extern "C" {

int ttemb_forward(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices,
                  const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                  const int32_t* nnz_dev, int64_t B, float* output, void* workspace,
                  int64_t workspace_bytes, void* plan, int64_t plan_bytes, void* stream) {
  ApiRange api_range("ttemb_forward");
  return forward_phase(0, shape, cores, indices, rowidx, offsets, nnz, nnz_dev, B, output, workspace, workspace_bytes, plan,
                       plan_bytes, stream);
}

int ttemb_forward_group(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices,
                        const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                        const int32_t* nnz_dev, int64_t B, float* output, void* workspace,
                        int64_t workspace_bytes, void* plan, int64_t plan_bytes, void* stream) {
  ApiRange api_range("ttemb_forward_group");
  return forward_phase(1, shape, cores, indices, rowidx, offsets, nnz, nnz_dev, B, output, workspace, workspace_bytes, plan,
                       plan_bytes, stream);
}

int ttemb_forward_lookup(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                         const int32_t* nnz_dev, int64_t B, float* output, void* workspace,
                         int64_t workspace_bytes, void* plan, int64_t plan_bytes, void* stream) {
  ApiRange api_range("ttemb_forward_lookup");
  return forward_phase(2, shape, cores, indices, rowidx, offsets, nnz, nnz_dev, B, output, workspace, workspace_bytes, plan,
                       plan_bytes, stream);
}

int ttemb_backward_dense(const ttemb_shape_t* shape, const float* const* cores,
                         const int64_t* indices, const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                         const int32_t* nnz_dev, int64_t B, const float* d_output,
                         float* const* d_cores, void* workspace, int64_t workspace_bytes,
                         const void* plan, int64_t plan_bytes, void* stream) {
  ApiRange api_range("ttemb_backward_dense");
  DevShape ds;
  int rc = pending_device_fault();
  if (rc) return rc;
  rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  rc = check_lookup_args(cores, indices, nnz, B);
  if (rc) return rc;
  if (d_cores == nullptr) return fail(TTEMB_E_BADARG, "d_cores is null");
  if (nnz > 0 && d_output == nullptr) return fail(TTEMB_E_BADARG, "d_output is null");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  CorePtrs cp;
  CorePtrsMut dp;
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) {
    cp.c[t] = t < ds.T ? cores[t] : nullptr;
    dp.c[t] = t < ds.T ? d_cores[t] : nullptr;
  }
  void* header = nullptr;   // the grouped path's persistent words: the first kFast3HeaderBytes of every lookup workspace
  if (workspace != nullptr && workspace_bytes >= kFast3HeaderBytes) {
    header = workspace;
    workspace = reinterpret_cast<char*>(workspace) + kFast3HeaderBytes;
    workspace_bytes -= kFast3HeaderBytes;
  }
  // the gradient scratch region behind the header is unused in dense mode
  const int64_t skip = grad_scratch_bytes(ds);
  char* ws = reinterpret_cast<char*>(workspace);
  int64_t rest = workspace_bytes > skip ? workspace_bytes - skip : 0;
  ws = ws ? ws + skip : nullptr;
  rc = resolve_rowidx(&rowidx, offsets, nnz, B, &ws, &rest, st, use_fast3(ds, nnz, B, offsets != nullptr));
  if (rc) return rc;
  return backward_into(ds, cp, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, dp, ws, rest, plan, plan_bytes, st, header);
}

static int fused_backward(const ttemb_shape_t* shape, float* const* cores, float* const* opt_state,
                          const int64_t* indices, const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                          const int32_t* nnz_dev, int64_t B, const float* d_output, float lr, float eps,
                          void* workspace, int64_t workspace_bytes, const void* plan, int64_t plan_bytes,
                          void* stream) {
  DevShape ds;
  int rc = pending_device_fault();
  if (rc) return rc;
  rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  rc = check_lookup_args(cores, indices, nnz, B);
  if (rc) return rc;
  if (nnz == 0) return TTEMB_OK;  // zero gradient: SGD is a no-op, Adagrad adds 0 and divides 0
  if (d_output == nullptr) return fail(TTEMB_E_BADARG, "d_output is null");
  const int64_t need = kFast3HeaderBytes + grad_scratch_bytes(ds);
  if (workspace == nullptr || workspace_bytes < need)
    return fail(TTEMB_E_WORKSPACE, "backward needs %lld workspace bytes, got %lld", (long long)need, (long long)workspace_bytes);
  void* header = workspace;   // the grouped path's persistent words: the first kFast3HeaderBytes of every lookup workspace
  workspace = reinterpret_cast<char*>(workspace) + kFast3HeaderBytes;
  workspace_bytes -= kFast3HeaderBytes;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  CorePtrs cp;
  CorePtrsMut gp;
  char* ws = reinterpret_cast<char*>(workspace);
  int64_t off = 0;
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) {
    cp.c[t] = t < ds.T ? cores[t] : nullptr;
    gp.c[t] = nullptr;
    if (t < ds.T) {
      gp.c[t] = reinterpret_cast<float*>(ws + off);
      off += align256((int64_t)ds.p[t] * ds.row_len[t] * 4);
    }
  }
  char* rest_ws = ws + off;
  int64_t rest = workspace_bytes - off;
  rc = resolve_rowidx(&rowidx, offsets, nnz, B, &rest_ws, &rest, st, use_fast3(ds, nnz, B, offsets != nullptr));
  if (rc) return rc;
  // the grouped path of a one-piece call applies the step inside its last kernel; every other route writes gradients, then steps
  const bool f3 = use_fast3(ds, nnz, B, offsets != nullptr) && one_piece(ds, nnz, B);
  bool aligned4 = true;
  FusedUpdate upd;
  memset(&upd, 0, sizeof(upd));
  for (int t = 0; t < ds.T; ++t) {
    upd.w[t] = cores[t];
    upd.st[t] = opt_state ? opt_state[t] : nullptr;
    aligned4 = aligned4 && cores[t] != nullptr && (!opt_state || opt_state[t] != nullptr);
  }
  if (!aligned4) return fail(TTEMB_E_BADARG, "null core / optimizer state");
  upd.lr = lr;
  upd.eps = eps;
  // the grouped path applies the step inside its last kernel; the generic path writes gradients, then steps
  bool grouped = false;
  rc = backward_into(ds, cp, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, gp, rest_ws, rest, plan, plan_bytes, st,
                     header, f3 ? &upd : nullptr, &grouped);
  if (rc || f3) return rc;
  Seg3 seg;
  memset(&seg, 0, sizeof(seg));
  int64_t nmax = 0;
  bool aligned = true;
  for (int t = 0; t < ds.T; ++t) {
    seg.w[t] = cores[t];
    seg.st[t] = opt_state ? opt_state[t] : nullptr;
    seg.g[t] = gp.c[t];
    seg.n[t] = (long long)ds.p[t] * ds.row_len[t];
    nmax = seg.n[t] > nmax ? seg.n[t] : nmax;
    aligned = aligned && ((reinterpret_cast<uintptr_t>(cores[t]) | (opt_state ? reinterpret_cast<uintptr_t>(opt_state[t]) : 0)) & 15) == 0;
  }
  if (!aligned) return fail(TTEMB_E_BADARG, "cores / optimizer state must be 16-byte aligned");
  int64_t blocks = (nmax / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(fused_step_kernel, dim3((unsigned)blocks, (unsigned)ds.T), dim3(256), 0, st, seg, lr, eps,
                     opt_state ? 1 : 0,
                     grouped ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(header) + kHeaderPoisonOffset) : nullptr);
  return check_hip(hipGetLastError(), "fused_step_kernel");
}

int ttemb_backward_sgd(const ttemb_shape_t* shape, float* const* cores, const int64_t* indices,
                       const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev, int64_t B,
                       const float* d_output, float lr, void* workspace, int64_t workspace_bytes,
                       const void* plan, int64_t plan_bytes, void* stream) {
  ApiRange api_range("ttemb_backward_sgd");
  return fused_backward(shape, cores, nullptr, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, lr, 0.f,
                        workspace, workspace_bytes, plan, plan_bytes, stream);
}

int ttemb_backward_adagrad(const ttemb_shape_t* shape, float* const* cores, float* const* opt_state,
                           const int64_t* indices, const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                           const int32_t* nnz_dev, int64_t B, const float* d_output, float lr,
                           float eps, void* workspace, int64_t workspace_bytes, const void* plan,
                           int64_t plan_bytes, void* stream) {
  ApiRange api_range("ttemb_backward_adagrad");
  if (opt_state == nullptr) return fail(TTEMB_E_BADARG, "opt_state is null");
  return fused_backward(shape, cores, opt_state, indices, rowidx, offsets, nnz, nnz_dev, B, d_output, lr, eps,
                        workspace, workspace_bytes, plan, plan_bytes, stream);
}

// ---- a window of a longer id list: one table of a table-batched call (include/ttemb.h) ----
int64_t ttemb_window_workspace_bytes(const ttemb_shape_t* shape, int32_t op, int64_t nnz, int64_t bags_total, int64_t B) {
  if (nnz < 0 || B < 0 || bags_total < B) return fail(TTEMB_E_BADARG, "negative size, or more bags in the window than in the call");
  if (op != TTEMB_OP_FORWARD && op != TTEMB_OP_BACKWARD) return fail(TTEMB_E_BADARG, "a window is looked up (TTEMB_OP_FORWARD) or differentiated (TTEMB_OP_BACKWARD)");
  DevShape ds;
  int rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  if (nnz == 0 || B == 0) return kFast3HeaderBytes;
  if (current_path() == TTEMB_PATH_GENERIC || current_path() == TTEMB_PATH_PER_BAG || !fast3_window_fits(ds, nnz, bags_total, B))
    return fail(TTEMB_E_UNSUPPORTED, "the grouped kernels do not cover this window (shape, size or forced path)");
  return kFast3HeaderBytes + fast3_window_workspace_bytes(ds, op == TTEMB_OP_BACKWARD, nnz);
}

static int window_args(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices, const int64_t* offsets,
                       int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, void** workspace, int64_t* workspace_bytes,
                       DevShape* ds, CorePtrs* cp, void** header) {
  int rc = pending_device_fault();
  if (rc) return rc;
  rc = make_dev_shape(shape, ds);
  if (rc) return rc;
  rc = check_lookup_args(cores, indices, nnz, B);
  if (rc) return rc;
  if (offsets == nullptr) return fail(TTEMB_E_BADARG, "a window needs the bag boundaries (offsets)");
  if (bag0 < 0 || B < 0 || bag0 + B > bags_total) return fail(TTEMB_E_BADARG, "the window [%lld, %lld) does not lie inside the call's %lld bags",
                                                                (long long)bag0, (long long)(bag0 + B), (long long)bags_total);
  if (current_path() == TTEMB_PATH_GENERIC || current_path() == TTEMB_PATH_PER_BAG)
    return fail(TTEMB_E_UNSUPPORTED, "a window is served by the grouped kernels (path forced elsewhere)");
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) cp->c[t] = t < ds->T ? cores[t] : nullptr;
  if (*workspace == nullptr || *workspace_bytes < kFast3HeaderBytes) return fail(TTEMB_E_WORKSPACE, "a window call needs ttemb_window_workspace_bytes() bytes");
  *header = *workspace;
  *workspace = reinterpret_cast<char*>(*workspace) + kFast3HeaderBytes;
  *workspace_bytes -= kFast3HeaderBytes;
  return TTEMB_OK;
}

int ttemb_forward_window(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices, const int64_t* offsets,
                         int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, float* output, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  ApiRange api_range("ttemb_forward_window");
  DevShape ds;
  CorePtrs cp;
  void* header = nullptr;
  int rc = window_args(shape, cores, indices, offsets, nnz, bags_total, bag0, B, &workspace, &workspace_bytes, &ds, &cp, &header);
  if (rc || B == 0) return rc;
  if (output == nullptr) return fail(TTEMB_E_BADARG, "output is null");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (nnz == 0) return launch_zero(output + bag0 * ds.D, (size_t)B * ds.D * 4, st, "zero the window's rows");
  return launch_forward_window_fast3(ds, cp, indices, offsets, nnz, bags_total, bag0, B, output, workspace, workspace_bytes, st, header);
}

static int backward_window(const ttemb_shape_t* shape, float* const* cores, float* const* opt_state, float* const* d_cores,
                           const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B,
                           const float* d_output, float lr, float eps, void* workspace, int64_t workspace_bytes, void* stream) {
  DevShape ds;
  CorePtrs cp;
  void* header = nullptr;
  int rc = window_args(shape, cores, indices, offsets, nnz, bags_total, bag0, B, &workspace, &workspace_bytes, &ds, &cp, &header);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  CorePtrsMut dp;
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) dp.c[t] = (d_cores != nullptr && t < ds.T) ? d_cores[t] : nullptr;
  if (d_cores != nullptr) {   // dense: every gradient is written whole
    for (int t = 0; t < ds.T; ++t)
      if (d_cores[t] == nullptr) return fail(TTEMB_E_BADARG, "d_cores[%d] is null", t);
    if (nnz == 0 || B == 0) return launch_zero_cores(ds, dp, st);
  } else if (nnz == 0 || B == 0) {
    return TTEMB_OK;   // zero gradient: SGD is a no-op, Adagrad adds 0 and divides 0
  }
  if (d_output == nullptr) return fail(TTEMB_E_BADARG, "d_output is null");
  FusedUpdate upd;
  memset(&upd, 0, sizeof(upd));
  if (d_cores == nullptr) {
    for (int t = 0; t < ds.T; ++t) {
      upd.w[t] = cores[t];
      upd.st[t] = opt_state ? opt_state[t] : nullptr;
      if (cores[t] == nullptr || (opt_state && opt_state[t] == nullptr)) return fail(TTEMB_E_BADARG, "null core / optimizer state");
    }
    upd.lr = lr;
    upd.eps = eps;
  }
  return launch_backward_window_fast3(ds, cp, indices, offsets, nnz, bags_total, bag0, B, d_output, dp, workspace, workspace_bytes, st,
                                      d_cores == nullptr ? &upd : nullptr, header);
}

int ttemb_backward_dense_window(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices, const int64_t* offsets,
                                int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, const float* d_output, float* const* d_cores,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  ApiRange api_range("ttemb_backward_dense_window");
  if (d_cores == nullptr) return fail(TTEMB_E_BADARG, "d_cores is null");
  return backward_window(shape, const_cast<float* const*>(cores), nullptr, d_cores, indices, offsets, nnz, bags_total, bag0, B, d_output, 0.f, 0.f,
                         workspace, workspace_bytes, stream);
}

int ttemb_backward_sgd_window(const ttemb_shape_t* shape, float* const* cores, const int64_t* indices, const int64_t* offsets,
                              int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, const float* d_output, float lr,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  ApiRange api_range("ttemb_backward_sgd_window");
  return backward_window(shape, cores, nullptr, nullptr, indices, offsets, nnz, bags_total, bag0, B, d_output, lr, 0.f, workspace,
                         workspace_bytes, stream);
}

int ttemb_backward_adagrad_window(const ttemb_shape_t* shape, float* const* cores, float* const* opt_state, const int64_t* indices,
                                  const int64_t* offsets, int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B,
                                  const float* d_output, float lr, float eps, void* workspace, int64_t workspace_bytes, void* stream) {
  ApiRange api_range("ttemb_backward_adagrad_window");
  if (opt_state == nullptr) return fail(TTEMB_E_BADARG, "opt_state is null");
  return backward_window(shape, cores, opt_state, nullptr, indices, offsets, nnz, bags_total, bag0, B, d_output, lr, eps, workspace,
                         workspace_bytes, stream);
}

int ttemb_sgd_step(float* weights, const float* grads, int64_t n, float lr, void* stream) {
  ApiRange api_range("ttemb_sgd_step");
  if (n > 0 && (weights == nullptr || grads == nullptr)) return fail(TTEMB_E_BADARG, "null buffer");
  return run_sgd(weights, grads, n, lr, reinterpret_cast<hipStream_t>(stream));
}

int ttemb_sgd_step_guarded(float* weights, const float* grads, int64_t n, float lr, const float* skip, void* stream) {
  ApiRange api_range("ttemb_sgd_step_guarded");
  if (n > 0 && (weights == nullptr || grads == nullptr)) return fail(TTEMB_E_BADARG, "null buffer");
  return run_sgd(weights, grads, n, lr, reinterpret_cast<hipStream_t>(stream), skip);
}

int ttemb_adagrad_step(float* weights, float* state, const float* grads, int64_t n, float lr,
                       float eps, void* stream) {
  ApiRange api_range("ttemb_adagrad_step");
  if (n > 0 && (weights == nullptr || grads == nullptr || state == nullptr)) return fail(TTEMB_E_BADARG, "null buffer");
  return run_adagrad(weights, state, grads, n, lr, eps, reinterpret_cast<hipStream_t>(stream));
}

static int cache_update(const int64_t* indices, int64_t nnz, int64_t* hashtbl, int64_t* cache_freq, int64_t H, bool one_sweep,
                        void* stream) {
  if (nnz < 0) return fail(TTEMB_E_BADARG, "negative nnz");
  if (nnz == 0) return TTEMB_OK;
  if (H <= 0 || H > 0x7fffffffll) return fail(TTEMB_E_BADARG, "hashtbl_size %lld out of range", (long long)H);
  if (!indices || !hashtbl || !cache_freq) return fail(TTEMB_E_BADARG, "null buffer");
  return launch_cache_update(indices, nnz, hashtbl, cache_freq, H, reinterpret_cast<hipStream_t>(stream), one_sweep);
}

int ttemb_cache_update(const int64_t* indices, int64_t nnz, int64_t* hashtbl, int64_t* cache_freq,
                       int64_t H, void* stream) {
  ApiRange api_range("ttemb_cache_update");
  return cache_update(indices, nnz, hashtbl, cache_freq, H, false, stream);
}

int ttemb_cache_update_one_sweep(const int64_t* indices, int64_t nnz, int64_t* hashtbl, int64_t* cache_freq,
                                 int64_t H, void* stream) {
  ApiRange api_range("ttemb_cache_update_one_sweep");
  return cache_update(indices, nnz, hashtbl, cache_freq, H, true, stream);
}

int ttemb_cache_populate(const ttemb_shape_t* shape, const float* const* cores, int64_t* hashtbl,
                         int64_t* cache_freq, int32_t* cache_state, int64_t H, float* cache_weight,
                         int64_t C, void* workspace, int64_t workspace_bytes, void* stream) {
  ApiRange api_range("ttemb_cache_populate");
  DevShape ds;
  int rc = make_dev_shape(shape, &ds);
  if (rc) return rc;
  if (H <= 0 || H > 0x7fffffffll) return fail(TTEMB_E_BADARG, "hashtbl_size %lld out of range", (long long)H);
  if (C < 0 || C > H) return fail(TTEMB_E_BADARG, "cache rows %lld must be within [0, hashtbl_size]", (long long)C);
  if (!cores || !hashtbl || !cache_freq || !cache_state || (C > 0 && !cache_weight) || !workspace)
    return fail(TTEMB_E_BADARG, "null buffer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t* sorted_keys = nullptr;
  if (workspace_bytes < kFast3HeaderBytes) return fail(TTEMB_E_WORKSPACE, "cache_populate: workspace smaller than its header");
  workspace = reinterpret_cast<char*>(workspace) + kFast3HeaderBytes;   // (the lookups' persistent words)
  workspace_bytes -= kFast3HeaderBytes;
  rc = launch_cache_populate_rank(hashtbl, cache_freq, cache_state, H, C, workspace, workspace_bytes,
                                  &sorted_keys, st);
  if (rc || C == 0) return rc;
  CorePtrs cp;
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) cp.c[t] = t < ds.T ? cores[t] : nullptr;
  // rows of the C hottest ids straight into cache_weight (reference: prefetch in chunks of 200)
  return launch_forward_generic(ds, cp, sorted_keys, nullptr, nullptr, C, C, nullptr, cache_weight, st);
}

static int preprocess_impl(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B, int32_t warmup,
                           int64_t* hashtbl, int64_t* cache_freq, const int32_t* cache_state, int64_t H,
                           int64_t* indices_out, int64_t* rowidx_out, int32_t* cache_loc_out, int32_t* nnz_tt_dev,
                           int32_t* dup_stamp, void* workspace, int64_t workspace_bytes, void* stream) {
  if (nnz < 0 || B < 0) return fail(TTEMB_E_BADARG, "negative size");
  if (nnz > 0x7fffffffll) return fail(TTEMB_E_BADARG, "nnz exceeds int32 range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool passthrough = warmup != 0 || H == 0;
  if (nnz == 0) return nnz_tt_dev ? launch_set_count(nnz_tt_dev, 0, st) : TTEMB_OK;
  if (!indices || !offsets || !rowidx_out) return fail(TTEMB_E_BADARG, "null buffer");
  if (passthrough) {
    int rc = launch_rowidx(offsets, B, nnz, rowidx_out, st);
    if (rc) return rc;
    if (indices_out != nullptr && indices_out != indices) {
      rc = check_hip(hipMemcpyAsync(indices_out, indices, (size_t)nnz * 8, hipMemcpyDeviceToDevice, st), "copy indices");
      if (rc) return rc;
    }
    return nnz_tt_dev ? launch_set_count(nnz_tt_dev, (int32_t)nnz, st) : TTEMB_OK;
  }
  if (H > 0x7fffffffll) return fail(TTEMB_E_BADARG, "hashtbl_size out of range");
  if (!hashtbl || !cache_state || !indices_out || !cache_loc_out || !nnz_tt_dev || !workspace)
    return fail(TTEMB_E_BADARG, "null buffer");
  if (indices_out == indices) return fail(TTEMB_E_BADARG, "partition cannot run in place");
  if (workspace_bytes < kFast3HeaderBytes) return fail(TTEMB_E_WORKSPACE, "preprocess: workspace smaller than its header");
  return launch_partition(indices, offsets, nnz, B, hashtbl, cache_freq, cache_state, H, indices_out, rowidx_out,
                          cache_loc_out, nnz_tt_dev, dup_stamp, reinterpret_cast<char*>(workspace) + kFast3HeaderBytes,
                          workspace_bytes - kFast3HeaderBytes, st);
}

int ttemb_preprocess(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B,
                     int32_t warmup, const int64_t* hashtbl, const int32_t* cache_state, int64_t H,
                     int64_t* indices_out, int64_t* rowidx_out, int32_t* cache_loc_out,
                     int32_t* nnz_tt_dev, int32_t* dup_stamp, int32_t epoch, void* workspace, int64_t workspace_bytes,
                     void* stream) {
  ApiRange api_range("ttemb_preprocess");
  (void)epoch;   // ABI 1 took a per-call epoch for the stamps; position stamps need none
  return preprocess_impl(indices, offsets, nnz, B, warmup, const_cast<int64_t*>(hashtbl), nullptr, cache_state, H, indices_out,
                         rowidx_out, cache_loc_out, nnz_tt_dev, dup_stamp, workspace, workspace_bytes, stream);
}

int ttemb_preprocess_update(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B, int64_t* hashtbl,
                            int64_t* cache_freq, const int32_t* cache_state, int64_t H, int64_t* indices_out,
                            int64_t* rowidx_out, int32_t* cache_loc_out, int32_t* nnz_tt_dev, int32_t* dup_stamp,
                            void* workspace, int64_t workspace_bytes, void* stream) {
  ApiRange api_range("ttemb_preprocess_update");
  if (H <= 0 || !hashtbl || !cache_freq) return fail(TTEMB_E_BADARG, "ttemb_preprocess_update needs the hash table and its counters");
  return preprocess_impl(indices, offsets, nnz, B, 0, hashtbl, cache_freq, cache_state, H, indices_out, rowidx_out,
                         cache_loc_out, nnz_tt_dev, dup_stamp, workspace, workspace_bytes, stream);
}

static int check_cache_args(const void* loc, const void* rowidx, int64_t start, int64_t nnz, int64_t D) {
  if (nnz < 0 || start < 0) return fail(TTEMB_E_BADARG, "negative size");
  if (D <= 0 || D % 4 != 0) return fail(TTEMB_E_BADARG, "embedding_dim %lld must be a positive multiple of 4", (long long)D);
  if (nnz > 0 && (!loc || !rowidx)) return fail(TTEMB_E_BADARG, "null buffer");
  return TTEMB_OK;
}

int ttemb_cache_forward(const int32_t* cache_loc, const int64_t* rowidx, const int64_t* offsets, int64_t start,
                        const int32_t* start_dev, int64_t nnz, const float* cache_weight, int64_t D,
                        float* output, void* stream) {
  ApiRange api_range("ttemb_cache_forward");
  int rc = check_cache_args(cache_loc, rowidx, start, nnz, D);
  if (rc) return rc;
  if (nnz > 0 && (!cache_weight || !output)) return fail(TTEMB_E_BADARG, "null buffer");
  return launch_cache_forward(cache_loc, rowidx, offsets, start, start_dev, nnz, cache_weight, D, output,
                              reinterpret_cast<hipStream_t>(stream));
}

int ttemb_cache_backward_sgd(const int32_t* cache_loc, const int64_t* rowidx, int64_t start,
                             const int32_t* start_dev, int64_t nnz, const float* d_output, int64_t D,
                             float lr, float* cache_weight, const int32_t* dup_dev, void* stream) {
  ApiRange api_range("ttemb_cache_backward_sgd");
  int rc = check_cache_args(cache_loc, rowidx, start, nnz, D);
  if (rc) return rc;
  if (nnz > 0 && (!cache_weight || !d_output)) return fail(TTEMB_E_BADARG, "null buffer");
  return launch_cache_scatter_add(cache_loc, rowidx, start, start_dev, nnz, d_output, D, -lr,
                                  cache_weight, dup_dev, reinterpret_cast<hipStream_t>(stream));
}

int ttemb_cache_backward_dense(const int32_t* cache_loc, const int64_t* rowidx, int64_t start,
                               const int32_t* start_dev, int64_t nnz, const float* d_output, int64_t D,
                               int64_t C, float* d_cache_weight, const int32_t* dup_dev, void* stream) {
  ApiRange api_range("ttemb_cache_backward_dense");
  int rc = check_cache_args(cache_loc, rowidx, start, nnz, D);
  if (rc) return rc;
  if (C < 0) return fail(TTEMB_E_BADARG, "negative cache rows");
  if (C > 0 && !d_cache_weight) return fail(TTEMB_E_BADARG, "null buffer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C > 0) {
    rc = launch_zero(d_cache_weight, (size_t)C * D * 4, st, "zero d_cache_weight");
    if (rc) return rc;
  }
  if (nnz > 0 && !d_output) return fail(TTEMB_E_BADARG, "null buffer");
  return launch_cache_scatter_add(cache_loc, rowidx, start, start_dev, nnz, d_output, D, 1.0f,
                                  d_cache_weight, dup_dev, st);
}

int ttemb_cache_backward_rowwise_adagrad(const int32_t* cache_loc, const int64_t* rowidx, int64_t start,
                                         const int32_t* start_dev, int64_t nnz, const float* d_output,
                                         int64_t D, float lr, float eps, float* cache_state_sum,
                                         float* cache_weight, void* stream) {
  ApiRange api_range("ttemb_cache_backward_rowwise_adagrad");
  int rc = check_cache_args(cache_loc, rowidx, start, nnz, D);
  if (rc) return rc;
  if (nnz > 0 && (!cache_weight || !d_output || !cache_state_sum)) return fail(TTEMB_E_BADARG, "null buffer");
  return launch_cache_rowwise_adagrad(cache_loc, rowidx, start, start_dev, nnz, d_output, D, lr, eps,
                                      cache_state_sum, cache_weight, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
