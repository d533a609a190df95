// C-ABI of the gfx950 spatial-VAE decoder (see include/svae.h): argument checks, workspace
// planning and the kernel launch sequences.  Nothing here allocates or synchronises.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <mutex>
#include <iterator>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "common.h"
#include <hip/hip_ext.h>
#include "dense.h"
#include "wgrad2.h"
#include "elementwise.h"
#include "encoder.h"
#include "split.h"

using namespace svae;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
    template <class T>
    T* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct Plan {
    // saved (forward -> backward)
    float* act[SVAE_MAX_HIDDEN + 1];
    // workspace
    float* wf[SVAE_MAX_HIDDEN];
    float* wb[SVAE_MAX_HIDDEN];
    float* tab;
    float4* posebuf;
    float* dh[2];
    float* do_p;
    float* slab;
    float* bslab;
    float* wpart;
    float* bpart;
    float* sgpart;
    float* sgimg;
    float* dcoords;
    float* sgtile;  // fused first-layer backward: per-tile sums (tiles x 2 x Hp x 4)
    float* dfpart;  // fused first-layer backward: per-column-block d(coords) (ntile x Mp x 2)
    float* llpart;  // fused Bernoulli finish: per-(image, pixel chunk) partial log-likelihoods (B x kFinishMaxChunks)
    uint4* splitA;  // fp16x3 mode: the row operand as hi/lo half fragments (Mp x Hp x 4 bytes)
    uint4* splitW;  // fp16x3 mode: one layer's weights as hi/lo half fragments
    uint4* savedC;    // fp16x3 mode, in `saved`: column fragments of a0 written by the coordinate layer (L == 2)
    uint4* splitC[2]; // fp16x3 mode: column fragments of the gradient and of a_{l-1} (weight gradient operands)
    float* hbpart;  // fp16x3 mode: partial column sums of dh_{L-1} from out_bwd_split_kernel (nparts x Hp)
    float* gscale;  // fp16x3 mode: {s, 1/s} power-of-two scale of the gradient entering the last hidden layer
    unsigned* amax; // fp16x3 mode: max |d loss / d logits| as float bits
    // split geometry
    int wg_nblk1, wg_S;
    long ob_oct_per_chunk;
    int ob_chunks;
    int l0_oct_per_chunk, l0_chunks_per_image;
    size_t saved_bytes, ws_bytes;
};

// SVAE_GEMM=fp16x3: hidden-layer GEMMs on the f16 matrix pipe with split (hi + lo/2048) operands, see split.h
std::mutex g_prof_mu;  // guards the profiler records and the mode word
// -1 = not chosen yet (first use takes SVAE_GEMM from the environment), 0 = fp32 MFMA, 1 = fp16x3; svae_gemm_mode_set()
int g_gemm_mode = -1;
bool split_mode() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_gemm_mode < 0) {
        const char* e = getenv("SVAE_GEMM");
        g_gemm_mode = (e && strcmp(e, "fp16x3") == 0) ? 1 : 0;
    }
    return g_gemm_mode == 1;
}

bool split_l0_on() {  // SVAE_SPLIT_L0=0: the coordinate layer writes fp32 only and conversion passes feed the GEMMs
    static const bool on = [] { const char* e = getenv("SVAE_SPLIT_L0"); return !(e && e[0] == '0'); }();
    return on;
}

// fp16x3, two hidden activations' worth of net (L == 2), plain coordinates, no residual: every consumer of a0 takes a
// fragment form (forward GEMM: rows; weight gradient and the fused first-layer epilogue: columns), so the coordinate layer
// skips the fp32 plane.  Forward and backward evaluate the same predicate.
bool split_a0_fragments_only(const Geo& g);

bool split_chain_on() {
    static const bool on = [] { const char* e = getenv("SVAE_SPLIT_CHAIN"); return !(e && e[0] == '0'); }();
    return on;
}

bool split_wgrad_on() {  // SVAE_SPLIT_WGRAD=0 keeps the fp32 weight-gradient kernel in fp16x3 mode
    static const bool on = [] { const char* e = getenv("SVAE_SPLIT_WGRAD"); return !(e && e[0] == '0'); }();
    return on;
}

bool split_a0_fragments_only(const Geo& g) {
    // opt-in (SVAE_SPLIT_A0=1): saves the 419 MB plane and 0.03 ms in the coordinate layer but costs the fused epilogue
    // 0.08 ms (8-byte strided fragment reads + conversions): a memory option, not a speed one
    static const bool on = [] { const char* e = getenv("SVAE_SPLIT_A0"); return e && e[0] == '1'; }();
    const char* fuse = getenv("SVAE_FUSE_OUT");
    return on && split_mode() && split_l0_on() && split_wgrad_on() && !(fuse && fuse[0] == '1') &&
           (g.act == SVAE_ACT_TANH || g.act == SVAE_ACT_SIGMOID) && g.ntile % 2 == 0 && g.L == 2 && g.in_dim == 2 &&
           !(g.flags & SVAE_FLAG_RESID);
}

// Output-layer backward.  SVAE_FUSE_OUT unset: the rank-1 fused form wherever it applies (one output channel, tanh / sigmoid,
// no residual, at least one hidden GEMM, fp32-MFMA mode) -- both GEMMs of the last hidden layer read a_{L-1} itself, no dh_{L-1}
// plane and no out_bwd pass -- and the streaming out_bwd pass elsewhere; "1": the generic fused form for any channel count
// (slower than the streaming pass on the fp32 MFMA: see wgrad_kernel); "0": always the streaming pass.
int fuse_out_mode() {
    const char* e = getenv("SVAE_FUSE_OUT");
    if (!e) return 2;
    return e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2);
}
bool split_active(const Geo& g) {
    return split_mode() && (g.act == SVAE_ACT_TANH || g.act == SVAE_ACT_SIGMOID) && g.ntile % 2 == 0;
}
// decided from the descriptor alone: the forward call packs the data-gradient weights of the last hidden layer with their
// rows scaled by w_o when (and only when) the backward call will take the rank-1 form
bool rank1_out(const Geo& g) {
    return fuse_out_mode() == 2 && !split_active(g) && g.C == 1 && g.L >= 2 && !(g.flags & SVAE_FLAG_RESID) &&
           (g.act == SVAE_ACT_TANH || g.act == SVAE_ACT_SIGMOID);
}

// What a forward call decided and baked into `saved` (rank-1 scaling of the packed data-gradient weights, fp16x3 operand
// forms, the a0-fragments-only layout), remembered per `saved` pointer so that the backward call can refuse a buffer planned
// under another GEMM mode or SVAE_FUSE_OUT setting instead of silently mixing the two (wrong gradients, no error).  A host-side
// table: the library never reads device memory back.  Bounded: beyond 4096 live buffers the oldest records are dropped (their
// backward calls then run unverified, as before).
struct PlanBits {
    unsigned bits;
    unsigned long long seq;
};
std::mutex g_plan_mu;
std::unordered_map<const void*, PlanBits> g_plans;
unsigned long long g_plan_seq = 0;
unsigned plan_bits(const Geo& g) {
    return (rank1_out(g) ? 1u : 0u) | (split_active(g) ? 2u : 0u) | ((unsigned)fuse_out_mode() << 2) |
           (split_a0_fragments_only(g) ? 16u : 0u) | ((split_mode() && split_l0_on()) ? 32u : 0u);
}
void remember_plan(const void* saved, unsigned bits) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (g_plans.size() >= 4096) {
        const unsigned long long cut = g_plan_seq - 2048;
        for (auto it = g_plans.begin(); it != g_plans.end();) it = it->second.seq < cut ? g_plans.erase(it) : std::next(it);
    }
    g_plans[saved] = PlanBits{bits, g_plan_seq++};
}
// -1: no record, else the recorded bits
long recorded_plan(const void* saved) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(saved);
    return it == g_plans.end() ? -1L : (long)it->second.bits;
}

// pixel chunks per image of logits_finish_bce_kernel: 1024 pixels per block, at most kFinishMaxChunks blocks per image
constexpr int kFinishMaxChunks = 64;
int finish_chunks(int N) {
    int c = (N + 1023) / 1024;
    return c < 1 ? 1 : (c > kFinishMaxChunks ? kFinishMaxChunks : c);
}

Plan make_plan(const Geo& g, void* saved, void* ws) {
    Plan p;
    const size_t MH = (size_t)g.Mp * g.Hp;
    Carver cs(saved);
    for (int l = 0; l < g.L; ++l) p.act[l] = cs.take<float>(MH);
    p.savedC = split_mode() ? cs.take<uint4>(MH / 4) : nullptr;  // fp16x3: column fragments of a0 for the weight gradient
    // what launch_prepare builds (packed weights, first-layer tables, poses: a few MB) also rides in `saved`, so the
    // backward call does not rebuild it; inference-only calls (saved == NULL) keep it in the workspace
    float* swf[SVAE_MAX_HIDDEN];
    float* swb[SVAE_MAX_HIDDEN];
    for (int l = 0; l + 1 < g.L; ++l) {
        swf[l] = cs.take<float>((size_t)g.Hp * g.Hp);
        swb[l] = cs.take<float>((size_t)g.Hp * g.Hp);
    }
    float* stab = cs.take<float>((size_t)g.B * g.Hp * kSlots);
    float4* sposebuf = cs.take<float4>(g.B);
    p.saved_bytes = (cs.off + 255) & ~size_t(255);

    p.wg_nblk1 = (g.ntile + 7) / 8;
    {
        const long nblk = (long)p.wg_nblk1 * p.wg_nblk1;
        long S = 256 / nblk;
        if (S < 1) S = 1;
        long cap = g.noct / 8;
        if (cap < 1) cap = 1;
        if (S > cap) S = cap;
        p.wg_S = (int)S;
    }
    {
        long rc = g.noct / 32;
        if (rc < 1) rc = 1;
        if (rc > 256) rc = 256;
        p.ob_oct_per_chunk = (g.noct + rc - 1) / rc;
        p.ob_chunks = (int)((g.noct + p.ob_oct_per_chunk - 1) / p.ob_oct_per_chunk);
    }
    {
        const int oimg = g.Npad / 8;
        const int xb = (g.Hp * 2 + 255) / 256;
        long want = (2048 + (long)g.B * xb - 1) / ((long)g.B * xb);
        if (want < 1) want = 1;
        int opc = (int)((oimg + want - 1) / want);
        if (opc < 8) opc = 8;
        if (opc > oimg) opc = oimg;
        p.l0_oct_per_chunk = opc;
        p.l0_chunks_per_image = (oimg + opc - 1) / opc;
    }

    Carver cw(ws);
    for (int l = 0; l + 1 < g.L; ++l) {
        p.wf[l] = cw.take<float>((size_t)g.Hp * g.Hp);
        p.wb[l] = cw.take<float>((size_t)g.Hp * g.Hp);
    }
    p.tab = cw.take<float>((size_t)g.B * g.Hp * kSlots);
    p.posebuf = cw.take<float4>(g.B);
    if (saved) {
        for (int l = 0; l + 1 < g.L; ++l) {
            p.wf[l] = swf[l];
            p.wb[l] = swb[l];
        }
        p.tab = stab;
        p.posebuf = sposebuf;
    }
    p.dh[0] = cw.take<float>(MH);
    p.dh[1] = cw.take<float>(MH);
    p.do_p = cw.take<float>((size_t)g.C * g.Mp);
    p.slab = cw.take<float>((size_t)p.wg_S * g.Hp * g.Hp);
    p.bslab = cw.take<float>((size_t)p.wg_S * 2 * g.Hp);
    size_t nparts = (size_t)(p.ob_chunks > p.wg_S ? p.ob_chunks : p.wg_S) * 2;  // out_bwd chunks or wgrad splits
    if (split_mode() && nparts < 256 * 4) nparts = 256 * 4;                        // out_bwd_split: up to 256 chunks x 4
    p.wpart = cw.take<float>(nparts * g.C * g.Hp);
    p.bpart = cw.take<float>(nparts * g.C);
    p.sgpart = cw.take<float>((size_t)g.B * p.l0_chunks_per_image * g.Hp * 2 * kSlots);
    p.sgimg = cw.take<float>((size_t)g.B * g.Hp * kSlots);
    p.dcoords = cw.take<float>((size_t)g.B * g.N * 2);
    p.sgtile = cw.take<float>((size_t)g.tiles * 2 * g.Hp * 4);
    p.dfpart = cw.take<float>((size_t)g.ntile * g.Mp * 2);
    p.llpart = cw.take<float>((size_t)g.B * kFinishMaxChunks);
    p.splitA = p.splitW = nullptr;
    p.splitC[0] = p.splitC[1] = nullptr;
    p.hbpart = nullptr;
    p.gscale = nullptr;
    p.amax = nullptr;
    if (split_mode()) {
        p.splitA = cw.take<uint4>(MH / 4);
        p.splitW = cw.take<uint4>((size_t)g.Hp * g.Hp / 4);
        p.splitC[0] = cw.take<uint4>(MH / 4);
        p.splitC[1] = cw.take<uint4>(MH / 4);
        p.hbpart = cw.take<float>((size_t)128 * 4 * g.Hp);
        p.gscale = cw.take<float>(64);
        p.amax = cw.take<unsigned>(64);
    }
    p.ws_bytes = (cw.off + 255) & ~size_t(255);
    return p;
}

int check_desc(const svae_desc* d) {
    if (!d) return fail(SVAE_E_INVALID, "null descriptor");
    if (d->B < 1 || d->N < 1 || d->H < 1) return fail(SVAE_E_INVALID, "B, N, H must be positive (got %d, %d, %d)", d->B, d->N, d->H);
    if (d->L < 1 || d->L - 1 > SVAE_MAX_HIDDEN) return fail(SVAE_E_INVALID, "num_layers %d outside 1..%d", d->L, SVAE_MAX_HIDDEN + 1);
    if (d->C < 1 || d->C > SVAE_MAX_OUT) return fail(SVAE_E_INVALID, "n_out %d outside 1..%d", d->C, SVAE_MAX_OUT);
    if (d->Zd < 0) return fail(SVAE_E_INVALID, "negative latent_dim");
    if (d->in_dim != 2 && d->in_dim != 5) return fail(SVAE_E_INVALID, "in_dim must be 2 or 5 (got %d)", d->in_dim);
    if (d->act < SVAE_ACT_TANH || d->act > SVAE_ACT_SIGMOID) return fail(SVAE_E_INVALID, "unknown activation %d", d->act);
    if ((d->flags & SVAE_FLAG_BILINEAR) && d->Zd == 0) return fail(SVAE_E_INVALID, "bilinear needs latent_dim > 0");
    if ((long)d->B * ((d->N + 31) / 32 * 32) > (1L << 30)) return fail(SVAE_E_INVALID, "B*N too large");
    return SVAE_OK;
}

int check_params(const svae_desc* d, const svae_params* p) {
    if (!p || !p->coord_w || !p->coord_b || !p->out_w || !p->out_b) return fail(SVAE_E_INVALID, "missing parameter pointer");
    if (d->Zd > 0 && !p->latent_w) return fail(SVAE_E_INVALID, "latent_dim > 0 but latent_w is null");
    if ((d->flags & SVAE_FLAG_BILINEAR) && !p->bilinear_w) return fail(SVAE_E_INVALID, "bilinear flag but bilinear_w is null");
    for (int l = 0; l + 1 < d->L; ++l)
        if (!p->hidden_w[l] || !p->hidden_b[l]) return fail(SVAE_E_INVALID, "hidden layer %d parameters are null", l);
    return SVAE_OK;
}

int check_pose(const svae_pose* pose) {
    if (!pose) return fail(SVAE_E_INVALID, "null pose");
    if (!pose->coords && !pose->grid) return fail(SVAE_E_INVALID, "pose needs coords or grid");
    return SVAE_OK;
}

int check_ws(const Plan& p, const void* ws, size_t ws_bytes) {
    if (!ws) return fail(SVAE_E_WORKSPACE, "null workspace");
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(SVAE_E_WORKSPACE, "workspace not 256-byte aligned");
    if (ws_bytes < p.ws_bytes) return fail(SVAE_E_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, p.ws_bytes);
    return SVAE_OK;
}

// ---- opt-in per-kernel timing with HIP events (svae_profile_*)
enum Kind { K_PREPARE = 0, K_LAYER0_FWD, K_DENSE_FWD, K_OUT_FWD, K_DLOGITS, K_OUT_BWD, K_WGRAD, K_WGRAD_REDUCE,
            K_DENSE_DGRAD, K_LAYER0_BWD, K_SMALL_BWD, K_BCE, K_GAUSSIAN, K_LATENT, K_ADAM, K_AUGMENT, K_ENCODER, K_COUNT };
const char* const kKindNames[SVAE_PROF_KINDS] = {"prepare", "layer0_fwd", "dense_fwd", "out_fwd", "dlogits", "out_bwd",
                                                 "wgrad", "wgrad_reduce", "dense_dgrad", "layer0_bwd", "small_bwd", "bce",
                                                 "gaussian", "latent", "adam", "augment", "encoder", "", "", ""};
static_assert(K_COUNT <= SVAE_PROF_KINDS, "svae_profile_read arrays too small");
// which kernel family a call actually dispatched (svae_path_counts): the GEMM mode is a request, the plan decides per
// geometry (fp16x3 falls back to the fp32 kernels for unbounded activations and odd tile counts), and a test must be able
// to tell a run of the split kernels from a silent fallback
enum Path { P_DENSE_FP32_FWD = 0, P_DENSE_FP32_DGRAD, P_WGRAD_FP32, P_DENSE_SPLIT_FWD, P_DENSE_SPLIT_DGRAD, P_WGRAD_SPLIT,
            P_OUT_BWD_STREAM, P_OUT_BWD_SPLIT, P_OUT_BWD_RANK1, P_OUT_BWD_FUSED_GENERIC, P_DENSE4, P_DENSE4_TAIL, P_WGRAD2, P_COUNT };
const char* const kPathNames[SVAE_PATH_KINDS] = {"dense_fp32_fwd", "dense_fp32_dgrad", "wgrad_fp32", "dense_split_fwd",
                                                  "dense_split_dgrad", "wgrad_split", "out_bwd_stream", "out_bwd_split",
                                                  "out_bwd_rank1", "out_bwd_fused_generic", "dense4", "dense4_tail", "wgrad2", "", "", ""};
static_assert(P_COUNT <= SVAE_PATH_KINDS, "svae_path_counts array too small");
std::atomic<long long> g_path[SVAE_PATH_KINDS];
inline void took(int path) { g_path[path].fetch_add(1, std::memory_order_relaxed); }

struct ProfRec { hipEvent_t a, b; int kind; };
int g_prof_level = 0;  // 0 off, 1 = the three MFMA GEMM kernels only, 2 = every kernel
std::vector<ProfRec> g_prof_used, g_prof_free;

// Two ways to time a launch.  (1) A Scope brackets whatever is launched during its lifetime with two hipEventRecord calls.
// Each record is a packet of its own in the queue, and the next kernel does not start before it has retired: ~3 us of idle
// chip per record, 0.02 ms per step at BASELINE cfg 2 with the three GEMMs bracketed (bench.py: 2.667 -> 2.649 ms per step
// with profiling off).  (2) `attach`: the Scope only reserves the event pair and the ONE kernel launched through launch_gemm
// during its lifetime carries it (hipExtLaunchKernelGGL: the dispatch packet's own start / stop timestamps): no extra packet,
// and the elapsed time is the kernel's, not the kernel's plus the records'.  The fp32 GEMM launches use (2).
struct Scope;
thread_local Scope* t_attach = nullptr;

struct Scope {
    hipStream_t st;
    ProfRec rec;
    bool on, attach, carried;
    Scope(int kind, hipStream_t s, bool attach_to_launch = false) : st(s), on(false), attach(attach_to_launch), carried(false) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_level == 0) return;
        if (g_prof_level == 1 && kind != K_DENSE_FWD && kind != K_DENSE_DGRAD && kind != K_WGRAD) return;
        if (!g_prof_free.empty()) {
            rec = g_prof_free.back();
            g_prof_free.pop_back();
        } else if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) {
            return;
        }
        rec.kind = kind;
        on = true;
        if (attach) t_attach = this;
        else (void)hipEventRecord(rec.a, st);
    }
    ~Scope() {
        if (!on) return;
        if (attach) t_attach = nullptr;
        else (void)hipEventRecord(rec.b, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!attach || carried) g_prof_used.push_back(rec);
        else g_prof_free.push_back(rec);   // nothing was launched through launch_gemm: the pair was never recorded
    }
};

// the launch of a GEMM kernel: carries the enclosing attach-Scope's events if there is one (the first launch only)
template <class K, class... Args>
inline void launch_gemm(K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t st, Args... args) {
    Scope* sc = t_attach;
    if (sc && !sc->carried) {
        sc->carried = true;
        hipExtLaunchKernelGGL(kernel, grid, block, lds, st, sc->rec.a, sc->rec.b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
    }
}

int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SVAE_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SVAE_OK;
}

inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

PoseArgs pose_args(const svae_pose* pose) {
    PoseArgs pa;
    pa.coords = pose->coords;
    pa.grid = pose->grid;
    pa.theta = pose->coords ? nullptr : pose->theta;
    pa.dx = pose->coords ? nullptr : pose->dx;
    return pa;
}

RowGeo row_geo(const Geo& g) {
    RowGeo r;
    r.N = g.N; r.Npad = g.Npad; r.Hp = g.Hp; r.in_dim = g.in_dim; r.act = g.act; r.B = g.B;
    return r;
}

// tables + packed weights, one launch (prepare_kernel): built by the forward call and carried to the backward call in `saved`
void launch_prepare(const Geo& g, const Plan& pl, const svae_params* p, const PoseArgs& pa, const float* z, hipStream_t st) {
    Scope prof(K_PREPARE, st);
    PrepareArgs a;
    a.coord_w = p->coord_w; a.coord_b = p->coord_b;
    a.latent_w = g.Zd > 0 ? p->latent_w : nullptr;
    a.bil_w = (g.flags & SVAE_FLAG_BILINEAR) ? p->bilinear_w : nullptr;
    a.z = z; a.tab = pl.tab; a.posebuf = pl.posebuf; a.pose = pa;
    a.B = g.B; a.H = g.H; a.Hp = g.Hp; a.Zd = g.Zd; a.in_dim = g.in_dim;
    const long nt = (long)g.B * g.Hp;
    a.nb_tab = (int)blocks_for(nt > g.B ? nt : g.B);
    a.nb_pack = (int)blocks_for((long)g.Hp * g.Hp);
    a.nlayers = g.L - 1;
    for (int l = 0; l < SVAE_MAX_HIDDEN; ++l) {
        const bool on = l + 1 < g.L;
        a.W[l] = on ? p->hidden_w[l] : nullptr;
        a.wf[l] = on ? pl.wf[l] : nullptr;
        a.wb[l] = on ? pl.wb[l] : nullptr;
        a.wb_row_scale[l] = (on && l == g.L - 2 && rank1_out(g)) ? p->out_w : nullptr;
    }
    hipLaunchKernelGGL(prepare_kernel, dim3((unsigned)(a.nb_tab + a.nb_pack * a.nlayers)), dim3(256), 0, st, a);
}

template <int NT, bool DGRAD, bool RESID, bool FIRST, int LASTD, int CF = 0>
void launch_dense_ntr(const DenseArgs& a, dim3 grid, hipStream_t st) {
    // weight buffers, plus the W_o table (max channels x max width) behind them for the generic LASTD form
    constexpr int kLds = DenseCfg<NT>::LDS_BYTES + (LASTD == 1 ? SVAE_MAX_OUT * 4096 * 4 : 0);
    const int lds = DenseCfg<NT>::LDS_BYTES + (LASTD == 1 ? SVAE_MAX_OUT * a.Hp * 4 : 0);
    static bool attr_set = false;  // LDS beyond 64 KiB needs the opt-in attribute once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_kernel<NT, DGRAD, RESID, FIRST, LASTD, CF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLds < 160 * 1024 ? kLds : 160 * 1024);
        attr_set = true;
    }
    launch_gemm(dense_kernel<NT, DGRAD, RESID, FIRST, LASTD, CF>, grid, dim3(256), (unsigned)lds, st, a);
}

template <int NT, bool DGRAD>
void launch_dense_nt(const DenseArgs& a, dim3 grid, hipStream_t st, bool first = false, int lastd = 0, int cf = 0) {
    if constexpr (!DGRAD && NT <= 4) {  // forward of the last hidden layer with the output layer's logits in the epilogue
        if (cf >= 1) {                  // any channel count (a.C): the kernel takes CF as a flag
            if (a.resid) launch_dense_ntr<NT, false, true, false, 0, 1>(a, grid, st);
            else launch_dense_ntr<NT, false, false, false, 0, 1>(a, grid, st);
            return;
        }
    }
    // The fused variants (FIRST epilogue / LASTD prologue) carry more live registers; at NT >= 8 hipcc starts to
    // re-home the in-flight registers of the asm loads (tools/check_asm_loads.py), so those instantiations do
    // not exist: dense_nt_first() caps the column tiles at 4 for them.
    if constexpr (DGRAD && NT <= 4) {
      if (first || lastd) {
        if (lastd == 2) {        // rank-1 form (tanh), never with a residual (checked by the caller)
            if (first) launch_dense_ntr<NT, true, false, true, 2>(a, grid, st);
            else launch_dense_ntr<NT, true, false, false, 2>(a, grid, st);
        } else if (lastd == 3) {  // rank-1 form (sigmoid)
            if (first) launch_dense_ntr<NT, true, false, true, 3>(a, grid, st);
            else launch_dense_ntr<NT, true, false, false, 3>(a, grid, st);
        } else if (lastd == 1) {  // generic form, never with a residual
            if (first) launch_dense_ntr<NT, true, false, true, 1>(a, grid, st);
            else launch_dense_ntr<NT, true, false, false, 1>(a, grid, st);
        } else {
            if (a.resid) launch_dense_ntr<NT, true, true, true, 0>(a, grid, st);
            else launch_dense_ntr<NT, true, false, true, 0>(a, grid, st);
        }
        return;
      }
    }
    if (a.resid) launch_dense_ntr<NT, DGRAD, true, false, 0>(a, grid, st);
    else launch_dense_ntr<NT, DGRAD, false, false, 0>(a, grid, st);
}

// column tiles accumulated per pass: the widest that divides the layer, unless SVAE_DENSE_NT caps it
int dense_nt_for(int ntile) {
    static const int cap = [] {
        const char* e = getenv("SVAE_DENSE_NT");
        const int v = e ? atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? v : 4;
    }();
    int nt = 16;
    while (nt > 1 && (ntile % nt != 0 || nt > cap)) nt >>= 1;
    return nt;
}

// the fused variants (first-layer epilogue, last-layer prologue) keep more values live: at most 4 column tiles
int dense_nt_first(int ntile) {
    int nt = dense_nt_for(ntile);
    while (nt > 4) nt >>= 1;
    return nt;
}

template <int NT, bool RESID>
void launch_split_fwd_c(const SplitArgs& a, dim3 grid, int cf, hipStream_t st) {
    constexpr int lds = SplitCfg<NT>::LDS_BYTES;
    switch (cf) {
        case 1: hipLaunchKernelGGL((dense_split_kernel<NT, 0, RESID, 1>), grid, dim3(kSplitWaves * 64), lds, st, a); break;
        case 2: hipLaunchKernelGGL((dense_split_kernel<NT, 0, RESID, 2>), grid, dim3(kSplitWaves * 64), lds, st, a); break;
        default: hipLaunchKernelGGL((dense_split_kernel<NT, 0, RESID, 0>), grid, dim3(kSplitWaves * 64), lds, st, a); break;
    }
}

template <int NT>
void launch_split_bwd_c(const SplitArgs& a, dim3 grid, bool first, bool resid, hipStream_t st) {
    constexpr int lds = SplitCfg<NT>::LDS_BYTES;
    if (first) {
        if (resid) hipLaunchKernelGGL((dense_split_kernel<NT, 2, true, 0>), grid, dim3(SplitWaves<2>::value * 64), lds, st, a);
        else hipLaunchKernelGGL((dense_split_kernel<NT, 2, false, 0>), grid, dim3(SplitWaves<2>::value * 64), lds, st, a);
    } else {
        if (resid) hipLaunchKernelGGL((dense_split_kernel<NT, 1, true, 0>), grid, dim3(kSplitWaves * 64), lds, st, a);
        else hipLaunchKernelGGL((dense_split_kernel<NT, 1, false, 0>), grid, dim3(kSplitWaves * 64), lds, st, a);
    }
}

// one weight matrix -> split fragments in pl.splitW, its scale {s, 1/s} in pl.gscale[2..3]
void split_weights(const Geo& g, const Plan& pl, const float* W, int transpose, hipStream_t st) {
    (void)hipMemsetAsync(pl.amax + 1, 0, sizeof(unsigned), st);
    hipLaunchKernelGGL(split_wamax_kernel, dim3(64), dim3(256), 0, st, W, (long)g.H * g.H, pl.amax + 1);
    hipLaunchKernelGGL(split_weights_kernel, dim3(blocks_for((long)(g.Hp / 16) * g.ntile * 64)), dim3(256), 0, st, W, pl.splitW,
                       g.H, g.Hp, transpose, (const unsigned*)(pl.amax + 1), pl.gscale + 2);
}

int split_nt(const Geo& g) { return g.ntile % 4 == 0 ? 4 : 2; }  // column tiles per pass (the path needs ntile even)
int split_nt_fwd(const Geo& g) {  // SVAE_SPLIT_NT8=1: 8 tiles per forward pass (half the row-operand traffic, but two waves
                                  // per SIMD instead of four: measured 0.33 vs 0.29 ms at cfg 2, so off by default)
    static const bool nt8 = [] { const char* e = getenv("SVAE_SPLIT_NT8"); return e && e[0] == '1'; }();
    return (nt8 && g.ntile % 8 == 0) ? 8 : split_nt(g);
}

dim3 split_grid(const Geo& g, int nt, int waves = kSplitWaves) {  // see dense_split_kernel: (xcd, column block, group / 8)
    const long groups = (g.tiles + waves - 1) / waves;
    return dim3((unsigned)(((groups + 7) / 8) * 8 * (g.ntile / nt)));
}

// forward hidden layer in fp16x3 mode: split the weights and the row operand, then the f16-MFMA GEMM
void launch_split_fwd(const Geo& g, const Plan& pl, const float* in, const float* W, const float* bias, float* out, bool resid,
                      int cf, const float* out_w, bool rows_ready, hipStream_t st) {
    took(P_DENSE_SPLIT_FWD);
    {
        Scope prof(K_PREPARE, st);
        split_weights(g, pl, W, 0, st);
        if (!rows_ready)  // deeper layers: the previous GEMM wrote fp32 only
            hipLaunchKernelGGL(split_rows_kernel, dim3(blocks_for(g.tiles * (g.Hp / 16) * 64)), dim3(256), 0, st, in,
                               pl.splitA, g.tiles, g.Hp, (const float*)nullptr);
    }
    Scope prof(K_DENSE_FWD, st);
    SplitArgs a;
    a.as = pl.splitA; a.ws = pl.splitW; a.out = out; a.bias = bias; a.resid = in;
    a.tiles = g.tiles; a.Hp = g.Hp; a.H = g.H; a.act = g.act;
    a.out_w = out_w; a.lpart = pl.dfpart; a.Mp = g.Mp;
    a.aux = nullptr; a.auxc = nullptr; a.scale = nullptr; a.wscale = pl.gscale + 2; a.amax_out = nullptr; a.posebuf = nullptr; a.tab = nullptr;
    a.sgtile = nullptr;
    a.N = g.N; a.Timg = g.Timg;
    const int nt = split_nt_fwd(g);
    const dim3 grid = split_grid(g, nt);
    if (nt == 8) resid ? launch_split_fwd_c<8, true>(a, grid, cf, st) : launch_split_fwd_c<8, false>(a, grid, cf, st);
    else if (nt == 4) resid ? launch_split_fwd_c<4, true>(a, grid, cf, st) : launch_split_fwd_c<4, false>(a, grid, cf, st);
    else resid ? launch_split_fwd_c<2, true>(a, grid, cf, st) : launch_split_fwd_c<2, false>(a, grid, cf, st);
}

// data gradient of the LAST hidden layer in fp16x3 mode: dh (fp32, scaled by pl.gscale) -> split rows, W^T -> split weights
void launch_split_dgrad(const Geo& g, const Plan& pl, const float* dh, const float* W, const float* aux, float* out, bool resid,
                        bool first, const PoseArgs& pa, bool rows_ready, hipStream_t st) {
    took(P_DENSE_SPLIT_DGRAD);
    {
        Scope prof(K_PREPARE, st);
        split_weights(g, pl, W, 1, st);
        if (!rows_ready)
            hipLaunchKernelGGL(split_rows_kernel, dim3(blocks_for(g.tiles * (g.Hp / 16) * 64)), dim3(256), 0, st, dh, pl.splitA,
                               g.tiles, g.Hp, (const float*)pl.gscale);
    }
    Scope prof(K_DENSE_DGRAD, st);
    SplitArgs a;
    a.as = pl.splitA; a.ws = pl.splitW; a.out = out; a.bias = nullptr; a.resid = dh;
    a.tiles = g.tiles; a.Hp = g.Hp; a.H = g.H; a.act = g.act;
    a.out_w = nullptr; a.lpart = nullptr; a.Mp = g.Mp;
    a.aux = aux; a.scale = pl.gscale; a.wscale = pl.gscale + 2; a.pose = pa;
    a.auxc = (first && split_a0_fragments_only(g)) ? pl.savedC : nullptr;
    a.amax_out = nullptr;
    if (!first) {  // the result is the next (lower) layer's gradient: track its largest entry for that layer's scale
        (void)hipMemsetAsync(pl.amax + 2, 0, sizeof(unsigned), st);
        a.amax_out = pl.amax + 2;
    } a.posebuf = pl.posebuf; a.tab = pl.tab; a.sgtile = pl.sgtile;
    a.dfpart = pl.dfpart; a.N = g.N; a.Timg = g.Timg;
    const int nt = split_nt(g);
    const dim3 grid = split_grid(g, nt, first ? SplitWaves<2>::value : SplitWaves<1>::value);
    if (nt == 4) launch_split_bwd_c<4>(a, grid, first, resid, st);
    else launch_split_bwd_c<2>(a, grid, first, resid, st);
    if (!first)  // scale of the plane just written, for the layer below (stream order: after the GEMM, before its consumers)
        hipLaunchKernelGGL(split_scale_amax_kernel, dim3(1), dim3(1), 0, st, (const unsigned*)(pl.amax + 2), pl.gscale);
}

// output layer backward in fp16x3 mode: dh straight into pl.splitA (rows) and pl.splitC[0] (columns); returns the number
// of partial sums per column it left in pl.wpart / pl.bpart
template <int ACT, int C>
int launch_out_bwd_split_ac(const Geo& g, const Plan& pl, const float* a, const svae_params* p, float* dh32, hipStream_t st) {
    Scope prof(K_OUT_BWD, st);
    long chunks = g.tiles < 128 ? g.tiles : 128;
    const long per = (g.tiles + chunks - 1) / chunks;
    chunks = (g.tiles + per - 1) / per;
    const dim3 grid((unsigned)(g.Hp / 64), (unsigned)chunks);
    if (dh32)
        hipLaunchKernelGGL((out_bwd_split_kernel<ACT, C, true>), grid, dim3(256), 0, st, a, pl.do_p, p->out_w, dh32, pl.splitA,
                           pl.splitC[0], pl.wpart, pl.bpart, pl.hbpart, (const float*)pl.gscale, g.H, g.Hp, (long)g.Mp, g.tiles,
                           per);
    else
        hipLaunchKernelGGL((out_bwd_split_kernel<ACT, C, false>), grid, dim3(256), 0, st, a, pl.do_p, p->out_w, dh32, pl.splitA,
                           pl.splitC[0], pl.wpart, pl.bpart, pl.hbpart, (const float*)pl.gscale, g.H, g.Hp, (long)g.Mp, g.tiles,
                           per);
    return (int)chunks * 4;
}
template <int ACT>
int launch_out_bwd_split_a(const Geo& g, const Plan& pl, const float* a, const svae_params* p, float* dh32, hipStream_t st) {
    took(P_OUT_BWD_SPLIT);
    switch (g.C) {
        case 1: return launch_out_bwd_split_ac<ACT, 1>(g, pl, a, p, dh32, st);
        case 2: return launch_out_bwd_split_ac<ACT, 2>(g, pl, a, p, dh32, st);
        case 3: return launch_out_bwd_split_ac<ACT, 3>(g, pl, a, p, dh32, st);
        default: return launch_out_bwd_split_ac<ACT, 4>(g, pl, a, p, dh32, st);
    }
}

// weight gradient of the LAST hidden layer in fp16x3 mode (operands converted to column fragments first)
void launch_split_wgrad(const Geo& g, const Plan& pl, const float* dh, const float* aprev, const uint4* aprev_cols,
                        bool cols_ready, hipStream_t st) {
    took(P_WGRAD_SPLIT);
    {
        Scope prof(K_PREPARE, st);
        if (!cols_ready)
            hipLaunchKernelGGL(split_cols_kernel, dim3(blocks_for(g.noct * g.Hp)), dim3(256), 0, st, dh, pl.splitC[0], g.noct,
                               g.Hp, (const float*)pl.gscale);
        if (!aprev_cols)  // deeper stacks: a_{L-2} came out of a GEMM epilogue in fp32 only
            hipLaunchKernelGGL(split_cols_kernel, dim3(blocks_for(g.noct * g.Hp)), dim3(256), 0, st, aprev, pl.splitC[1],
                               g.noct, g.Hp, (const float*)nullptr);
    }
    Scope prof(K_WGRAD, st);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_wgrad_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kSplitWgradLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_wgrad_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kSplitWgradLds);
        attr_set = true;
    }
    SplitWgradArgs w;
    w.dh = pl.splitC[0]; w.ap = aprev_cols ? aprev_cols : pl.splitC[1]; w.slab = pl.slab; w.bslab = pl.bslab; w.gscale = pl.gscale;
    w.nsteps = (long)g.Mp / 16; w.Hp = g.Hp; w.nblk1 = pl.wg_nblk1; w.S = pl.wg_S;
    const dim3 grid(pl.wg_nblk1 * pl.wg_nblk1 * pl.wg_S);
    if (cols_ready) hipLaunchKernelGGL(split_wgrad_kernel<false>, grid, dim3(256), kSplitWgradLds, st, w);  // db: out_bwd_split
    else hipLaunchKernelGGL(split_wgrad_kernel<true>, grid, dim3(256), kSplitWgradLds, st, w);
}

// dense4_kernel (four row tiles per wave, NT = 2 column tiles: dense4.h) takes a hidden-layer GEMM when the row space is
// whole 128-row groups, the width whole 64-column blocks, there is no residual and no generic (LASTD == 1) output-layer form,
// and the launch is large enough for its coarser work quantum (128 rows x 64 columns x K per wave; measured, dense4.h):
// at least four rounds of the 512 resident workgroups.  SVAE_DENSE4=0 never, =1 whenever legal.
// Returns the column tiles per workgroup (2 or 1) dense4_kernel should run with, or 0 for dense_kernel.  NT = 2 (64-column
// blocks, two waves per SIMD) is the fast form but its work quantum is 128 rows x 64 columns x K per wave: below four rounds
// of the 512 resident workgroups (BASELINE cfg 1: 51 200 rows) the half-size quantum of NT = 1 (three waves per SIMD) balances
// the chip better (measured: tools/dense4_proto.hip, 0.247 vs 0.234 ms against dense_kernel's 0.240).  SVAE_DENSE4=0 never,
// =1 / =2: that NT whenever legal.
int use_dense4(const Geo& g, int resid, int lastd) {
    const char* e = getenv("SVAE_DENSE4");  // read per call, like SVAE_FUSE_OUT: a test can compare the kernels in one process
    const int mode = e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 3;
    if (mode == 0 || g.tiles % 4 != 0 || resid || lastd == 1) return 0;
    const bool nt2_ok = g.ntile % 2 == 0;
    if (mode == 2) return nt2_ok ? 2 : 1;
    if (mode == 1) return 1;
    const long wgs2 = ((g.tiles / 4 + 3) / 4) * (g.ntile / 2);
    return (nt2_ok && wgs2 >= 2048) ? 2 : 1;
}

template <int NT, bool DGRAD, bool FIRST, int LASTD, int CF>
void launch_dense4_v(const DenseArgs& a, long groups, long set0, long nsets, hipStream_t st) {
    const dim3 grid((unsigned)(((nsets + 7) / 8) * 8 * ((a.Hp / 32) / NT)));
    // LDS: the two weight buffers, and for FIRST 3 x 128 floats per wave of per-row operands behind them
    constexpr int lds = DenseCfg<NT>::LDS_BYTES + (FIRST ? 4 * 3 * 128 * 4 : 0);
    // the grid is padded to whole XCD octets of sets: row groups beyond this launch's sets are dead in it (they belong to the
    // other launch of a split layer, or do not exist)
    const long gend = (set0 + nsets) * 4 < groups ? (set0 + nsets) * 4 : groups;
    launch_gemm(dense4_kernel<NT, DGRAD, FIRST, LASTD, CF>, grid, dim3(256), (unsigned)lds, st, a, gend, set0);
}

// How a hidden-layer GEMM was launched: column tiles per workgroup (its partial results -- a.lpart, a.dfpart -- come per column
// block of 32 x that many columns), and, when the layer was split in two launches, the first padded row of the half-width
// tail launch and its column tiles per workgroup (m_split = Mp: one launch).
struct DenseBlocks {
    int nt;
    long m_split;
    int nt_tail;
};

// The last partial round of a dense4_kernel<2> launch runs one workgroup per CU, and a wave's unit (128 rows x 64 columns x K)
// is serial on its SIMD: measured at BASELINE cfg 2, 3200 workgroups take 8.0 % longer than 3072 for 4.2 % more rows
// (profiles/r03_dense4_tail.txt).  When the launch is a few rounds long and its last round is less than half full, the sets
// of that round run as a second launch at half the block width (dense4_kernel<1>: twice the workgroups, half the unit).
// Returns the number of sets the wide launch keeps (all of them: no split).  tail_ok: the partial buffers have room for the
// tail's 32-column blocks (C <= 2 for the logits partials).
long dense4_main_sets(const Geo& g, long sets, bool tail_ok) {
    const char* e = getenv("SVAE_DENSE4_TAIL");   // "0": never split; "=<k>": the wide launch keeps k sets (tests)
    if ((e && e[0] == '0') || !tail_ok) return sets;
    if (e && e[0] == '=') {
        const long k = atol(e + 1);
        return (k > 0 && k < sets) ? k : sets;
    }
    static const long slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return (long)cus * 2;   // two resident workgroups of dense4_kernel<2> per CU
    }();
    const long nblk = g.ntile / 2;
    const long W = sets * nblk;
    if (slots % nblk != 0) return sets;
    const long full = (W / slots) * slots, rest = W - full;
    if (full < 2 * slots || rest == 0 || W >= 16 * slots || 20 * rest > 9 * slots) return sets;   // 2..16 rounds, last <= 0.45 full
    return full / nblk;
}

// wide workgroups over the sets [0, main_sets), half-width ones over [main_sets, sets): one launch (dense4_dual_kernel)
template <bool DGRAD, bool FIRST, int LASTD, int CF>
void launch_dense4_dual(const DenseArgs& a, long groups, long sets, long main_sets, hipStream_t st) {
    const unsigned ntile = (unsigned)(a.Hp / 32);
    const unsigned grid_main = (unsigned)(((main_sets + 7) / 8) * 8) * (ntile / 2);
    const unsigned grid_tail = (unsigned)(((sets - main_sets + 7) / 8) * 8) * ntile;
    constexpr int lds = DenseCfg<2>::LDS_BYTES + (FIRST ? 4 * 3 * 128 * 4 : 0);
    launch_gemm(dense4_dual_kernel<DGRAD, FIRST, LASTD, CF>, dim3(grid_main + grid_tail), dim3(256), (unsigned)lds, st, a, groups,
                main_sets, grid_main);
}

template <bool DGRAD>
DenseBlocks launch_dense(const Geo& g, const DenseArgs& a, hipStream_t st, bool first = false, int lastd = 0, int cf = 0) {
    took(DGRAD ? P_DENSE_FP32_DGRAD : P_DENSE_FP32_FWD);
    if (lastd) took(lastd == 1 ? P_OUT_BWD_FUSED_GENERIC : P_OUT_BWD_RANK1);
    Scope prof(DGRAD ? K_DENSE_DGRAD : K_DENSE_FWD, st, true);
    if (const int nt4 = use_dense4(g, a.resid, lastd)) {
        took(P_DENSE4);
        const long groups = g.tiles / 4;
        const long sets = (groups + 3) / 4;
        auto go = [&](auto nt_tag, long set0, long nsets) {
            constexpr int NT = decltype(nt_tag)::value;
            if constexpr (!DGRAD) {
                switch (cf) {
                    case 0: launch_dense4_v<NT, false, false, 0, 0>(a, groups, set0, nsets, st); break;
                    case 1: launch_dense4_v<NT, false, false, 0, 1>(a, groups, set0, nsets, st); break;
                    case 2: launch_dense4_v<NT, false, false, 0, 2>(a, groups, set0, nsets, st); break;
                    case 3: launch_dense4_v<NT, false, false, 0, 3>(a, groups, set0, nsets, st); break;
                    default: launch_dense4_v<NT, false, false, 0, 4>(a, groups, set0, nsets, st); break;
                }
            } else {
                if (first) {
                    if (lastd == 2) launch_dense4_v<NT, true, true, 2, 0>(a, groups, set0, nsets, st);
                    else if (lastd == 3) launch_dense4_v<NT, true, true, 3, 0>(a, groups, set0, nsets, st);
                    else launch_dense4_v<NT, true, true, 0, 0>(a, groups, set0, nsets, st);
                } else {
                    if (lastd == 2) launch_dense4_v<NT, true, false, 2, 0>(a, groups, set0, nsets, st);
                    else if (lastd == 3) launch_dense4_v<NT, true, false, 3, 0>(a, groups, set0, nsets, st);
                    else launch_dense4_v<NT, true, false, 0, 0>(a, groups, set0, nsets, st);
                }
            }
        };
        if (nt4 == 1) {
            go(std::integral_constant<int, 1>(), 0, sets);
            return DenseBlocks{1, (long)g.Mp, 1};
        }
        if (cf > 2) {   // three or four output channels: the logits partials have no room for 32-column blocks
            if constexpr (!DGRAD) {
                if (cf == 3) launch_dense4_v<2, false, false, 0, 3>(a, groups, 0, sets, st);
                else launch_dense4_v<2, false, false, 0, 4>(a, groups, 0, sets, st);
            }
            return DenseBlocks{2, (long)g.Mp, 1};
        }
        const long main_sets = dense4_main_sets(g, sets, true);
        if (main_sets < sets) took(P_DENSE4_TAIL);
        if constexpr (!DGRAD) {
            switch (cf) {
                case 0: launch_dense4_dual<false, false, 0, 0>(a, groups, sets, main_sets, st); break;
                case 1: launch_dense4_dual<false, false, 0, 1>(a, groups, sets, main_sets, st); break;
                default: launch_dense4_dual<false, false, 0, 2>(a, groups, sets, main_sets, st); break;
            }
        } else {
            if (first) {
                if (lastd == 2) launch_dense4_dual<true, true, 2, 0>(a, groups, sets, main_sets, st);
                else if (lastd == 3) launch_dense4_dual<true, true, 3, 0>(a, groups, sets, main_sets, st);
                else launch_dense4_dual<true, true, 0, 0>(a, groups, sets, main_sets, st);
            } else {
                if (lastd == 2) launch_dense4_dual<true, false, 2, 0>(a, groups, sets, main_sets, st);
                else if (lastd == 3) launch_dense4_dual<true, false, 3, 0>(a, groups, sets, main_sets, st);
                else launch_dense4_dual<true, false, 0, 0>(a, groups, sets, main_sets, st);
            }
        }
        return DenseBlocks{2, main_sets < sets ? main_sets * 512 : (long)g.Mp, 1};
    }
    const int nt = (first || lastd || cf) ? dense_nt_first(g.ntile) : dense_nt_for(g.ntile);
    static const bool xcd_grid = [] { const char* e = getenv("SVAE_XCD_GRID"); return !(e && e[0] == '0'); }();
    const long groups = (g.tiles + 3) / 4;
    const dim3 grid = (xcd_grid && g.ntile / nt > 1) ? dim3((unsigned)(((groups + 7) / 8) * 8 * (g.ntile / nt)))
                                                     : dim3((unsigned)groups, (unsigned)(g.ntile / nt));
    switch (nt) {
        case 16: launch_dense_nt<16, DGRAD>(a, grid, st, first, lastd, cf); break;
        case 8: launch_dense_nt<8, DGRAD>(a, grid, st, first, lastd, cf); break;
        case 4: launch_dense_nt<4, DGRAD>(a, grid, st, first, lastd, cf); break;
        case 2: launch_dense_nt<2, DGRAD>(a, grid, st, first, lastd, cf); break;
        default: launch_dense_nt<1, DGRAD>(a, grid, st, first, lastd, cf); break;
    }
    return DenseBlocks{nt, (long)g.Mp, nt};
}

template <int CL, int R1 = 0>
void launch_wgrad_c(const WgradArgs& w, dim3 grid, hipStream_t st) {
    static bool attr_set = false;  // 144 KiB of LDS (4 waves x 4 ring slots x 9 KiB) needs the opt-in
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<CL, R1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kWgradLdsBytes);
        attr_set = true;
    }
    launch_gemm(wgrad_kernel<CL, R1>, grid, dim3(256), (unsigned)kWgradLdsBytes, st, w);
}

// SVAE_TAIL_MERGE=0: the first hidden layer's split-K reduction as a launch of its own (read per call, for A/B runs and tests)
bool tail_merge_on() {
    const char* e = getenv("SVAE_TAIL_MERGE");
    return !(e && e[0] == '0');
}

// wgrad2_kernel (two waves per SIMD, operands straight into registers) where it has the form: the plain weight gradient and
// the rank-1 LASTW forms.  SVAE_WGRAD2=0 keeps wgrad_kernel everywhere (read per call: tests compare the two in one process).
bool use_wgrad2(int cl, int r1) {
    const char* e = getenv("SVAE_WGRAD2");
    if (e && e[0] == '0') return false;
    return cl == 0 || r1 != 0;
}

// cl > 0: the LASTW forms (dh formed from a_{L-1} in registers); r1: the rank-1 form (cl == 1), 1 = tanh, 2 = sigmoid.
// nsplit row-range splits (the plan's wg_S: slab / bslab / wpart / bpart are sized for it); xcd: 1-D XCD-aware grid if it fits
void launch_wgrad(WgradArgs w, int nsplit, bool xcd, int cl, int r1, hipStream_t st) {
    took(P_WGRAD_FP32);
    Scope prof(K_WGRAD, st, true);
    const int ntile = w.Hp / 32;
    if (use_wgrad2(cl, r1)) {
        took(P_WGRAD2);
        const int nb2 = ((ntile + 3) / 4) * ((ntile + 7) / 8);
        w.S = (xcd && nb2 > 1 && nsplit % 8 == 0) ? nsplit : 0;
        const dim3 grid = w.S ? dim3(nb2 * nsplit) : dim3(nb2, nsplit);
        if (r1 == 1) launch_gemm(wgrad2_kernel<1>, grid, dim3(256), 0u, st, w);
        else if (r1 == 2) launch_gemm(wgrad2_kernel<2>, grid, dim3(256), 0u, st, w);
        else launch_gemm(wgrad2_kernel<0>, grid, dim3(256), 0u, st, w);
        return;
    }
    const int nb2 = w.nblk1 * w.nblk1;
    w.S = (xcd && nb2 > 1 && nsplit % 8 == 0) ? nsplit : 0;
    const dim3 grid = w.S ? dim3(nb2 * nsplit) : dim3(nb2, nsplit);
    if (r1 == 1) {
        launch_wgrad_c<1, 1>(w, grid, st);
        return;
    }
    if (r1 == 2) {
        launch_wgrad_c<1, 2>(w, grid, st);
        return;
    }
    switch (cl) {
        case 1: launch_wgrad_c<1>(w, grid, st); break;
        case 2: launch_wgrad_c<2>(w, grid, st); break;
        case 3: launch_wgrad_c<3>(w, grid, st); break;
        case 4: launch_wgrad_c<4>(w, grid, st); break;
        default: launch_wgrad_c<0>(w, grid, st); break;
    }
}

template <int ACT>
void launch_layer0_fwd(const Geo& g, const Plan& pl, const PoseArgs& pa, float* a0, hipStream_t st) {
    Scope prof(K_LAYER0_FWD, st);
    const int oimg = g.Npad / 8;
    const int cpi = (oimg + kL0Chunk - 1) / kL0Chunk;   // chunks per image, evenly sized
    const int opc = (oimg + cpi - 1) / cpi;
    const long nchunks = (long)g.B * cpi;
    const unsigned gy = (unsigned)(nchunks < 32768 ? nchunks : 32768);
    const unsigned gz = (unsigned)((nchunks + gy - 1) / gy);
    hipLaunchKernelGGL((layer0_fwd_kernel<ACT>), dim3(blocks_for(g.Hp * 2), gy, gz), dim3(256), 0, st, pa, pl.posebuf,
                       pl.tab, a0, row_geo(g), opc, (oimg + opc - 1) / opc, nchunks);
}

template <int ACT>
void launch_layer0_fwd_split(const Geo& g, const Plan& pl, const PoseArgs& pa, float* a0, bool saved_ok, hipStream_t st) {
    Scope prof(K_LAYER0_FWD, st);  // Hp is a multiple of 64 here (ntile even)
    hipLaunchKernelGGL((layer0_fwd_split_kernel<ACT>), dim3((unsigned)g.tiles, (unsigned)(g.Hp / 64)), dim3(256), 0, st, pa,
                       pl.posebuf, pl.tab, split_a0_fragments_only(g) ? (float*)nullptr : a0, pl.splitA,
                       (saved_ok && g.L == 2) ? pl.savedC : (uint4*)nullptr, row_geo(g));
}

template <int C>
void launch_out_fwd(const Geo& g, const float* a, const svae_params* p, float* y, float* logits, hipStream_t st) {
    Scope prof(K_OUT_FWD, st);
    long nb = (g.noct + 3) / 4;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL((out_fwd_kernel<C>), dim3((unsigned)nb), dim3(256), 0, st, a, p->out_w, p->out_b, y, logits,
                       row_geo(g), g.H, (g.flags & SVAE_FLAG_SOFTPLUS) ? 1 : 0, g.noct);
}

template <int ACT, int C>
void launch_out_bwd_ac(const Geo& g, const Plan& pl, const float* a, const svae_params* p, float* dh, hipStream_t st) {
    Scope prof(K_OUT_BWD, st);
    hipLaunchKernelGGL((out_bwd_kernel<ACT, C>), dim3(blocks_for(g.Hp * 2), pl.ob_chunks), dim3(256), 0, st, a, pl.do_p,
                       p->out_w, dh, pl.wpart, pl.bpart, g.H, g.Hp, (long)g.Mp, g.noct, pl.ob_oct_per_chunk);
}
template <int ACT>
void launch_out_bwd_a(const Geo& g, const Plan& pl, const float* a, const svae_params* p, float* dh, hipStream_t st) {
    took(P_OUT_BWD_STREAM);
    switch (g.C) {
        case 1: launch_out_bwd_ac<ACT, 1>(g, pl, a, p, dh, st); break;
        case 2: launch_out_bwd_ac<ACT, 2>(g, pl, a, p, dh, st); break;
        case 3: launch_out_bwd_ac<ACT, 3>(g, pl, a, p, dh, st); break;
        default: launch_out_bwd_ac<ACT, 4>(g, pl, a, p, dh, st); break;
    }
}

}  // namespace

extern "C" {

int svae_abi_version(void) { return SVAE_ABI_VERSION; }
const char* svae_last_error(void) { return g_err; }

size_t svae_saved_bytes(const svae_desc* d) {
    if (check_desc(d) != SVAE_OK) return 0;
    return make_plan(make_geo(*d), nullptr, nullptr).saved_bytes;
}

size_t svae_workspace_bytes(const svae_desc* d) {
    if (check_desc(d) != SVAE_OK) return 0;
    return make_plan(make_geo(*d), nullptr, nullptr).ws_bytes;
}

}  // extern "C"

namespace {
// svae_decoder_forward, optionally with the Bernoulli log-likelihood folded into the output layer (bce_target != NULL)
int decoder_forward_impl(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z, float* y,
                         float* logits, void* saved, void* ws, size_t ws_bytes, svae_stream_t stream, const float* bce_target,
                         float* loglik, float* dll_dy) {
    int rc;
    if ((rc = check_desc(d)) || (rc = check_params(d, p)) || (rc = check_pose(pose))) return rc;
    if (!y) return fail(SVAE_E_INVALID, "y is null");
    if (bce_target && !loglik) return fail(SVAE_E_INVALID, "a Bernoulli target needs the loglik output");
    if (bce_target && (d->flags & SVAE_FLAG_SOFTPLUS))
        return fail(SVAE_E_INVALID, "softplus output with a Bernoulli likelihood: the reference's binary_cross_entropy rejects "
                                    "values above 1 (train_mnist.py:81)");
    if (d->Zd > 0 && !z) return fail(SVAE_E_INVALID, "latent_dim > 0 but z is null");
    if (saved && (reinterpret_cast<uintptr_t>(saved) & 255)) return fail(SVAE_E_WORKSPACE, "saved not 256-byte aligned");
    const Geo g = make_geo(*d);
    Plan pl = make_plan(g, saved, ws);
    if ((rc = check_ws(pl, ws, ws_bytes))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PoseArgs pa = pose_args(pose);
    if (!saved) {  // inference only: ping-pong through the (otherwise unused) gradient buffers
        for (int l = 0; l < g.L; ++l) pl.act[l] = pl.dh[l & 1];
    } else {
        remember_plan(saved, plan_bits(g));
    }

    // the output layer's logits come out of the last hidden layer's epilogue (no second pass over a_{L-1}) whenever that
    // layer is a dense_kernel launch; SVAE_FUSE_LOGITS=0 keeps the separate out_fwd pass.  The per-column-block partials
    // [nblk][C][Mp] live in the (forward-idle) dfpart area of ntile * 2 * Mp floats: C <= 2 NT; the fp16x3 kernels carry
    // at most two channels.
    const char* fuse_e = getenv("SVAE_FUSE_LOGITS");  // read per call (a test switches it)
    const bool fuse_env = !(fuse_e && fuse_e[0] == '0');
    const bool split_fwd = split_mode() && (g.act == SVAE_ACT_TANH || g.act == SVAE_ACT_SIGMOID) && g.ntile % 2 == 0;
    const bool fuse_logits = fuse_env && g.L >= 2 &&
                             (split_fwd ? g.C <= 2
                                        : g.C <= 2 * (use_dense4(g, g.flags & SVAE_FLAG_RESID, 0) ? use_dense4(g, g.flags & SVAE_FLAG_RESID, 0)
                                                                                                   : dense_nt_first(g.ntile)));

    launch_prepare(g, pl, p, pa, z, st);
    // fp16x3: bounded operands only, contraction length a multiple of 64
    const bool split = split_mode() && (g.act == SVAE_ACT_TANH || g.act == SVAE_ACT_SIGMOID) && g.ntile % 2 == 0;
    const bool split_l0 = split_l0_on() && split && g.L >= 2;
    if (split_l0) {  // a0 leaves the coordinate layer in both forms: no conversion pass before the first GEMM
        if (g.act == SVAE_ACT_TANH) launch_layer0_fwd_split<SVAE_ACT_TANH>(g, pl, pa, pl.act[0], saved != nullptr, st);
        else launch_layer0_fwd_split<SVAE_ACT_SIGMOID>(g, pl, pa, pl.act[0], saved != nullptr, st);
    } else {
        switch (g.act) {
            case SVAE_ACT_TANH: launch_layer0_fwd<SVAE_ACT_TANH>(g, pl, pa, pl.act[0], st); break;
            case SVAE_ACT_LEAKYRELU: launch_layer0_fwd<SVAE_ACT_LEAKYRELU>(g, pl, pa, pl.act[0], st); break;
            case SVAE_ACT_RELU: launch_layer0_fwd<SVAE_ACT_RELU>(g, pl, pa, pl.act[0], st); break;
            default: launch_layer0_fwd<SVAE_ACT_SIGMOID>(g, pl, pa, pl.act[0], st); break;
        }
    }
    DenseBlocks cfb{dense_nt_first(g.ntile), (long)g.Mp, 1};  // how the launch that wrote the partial logits was blocked
    for (int l = 1; l < g.L; ++l) {
        if (split) {
            launch_split_fwd(g, pl, pl.act[l - 1], p->hidden_w[l - 1], p->hidden_b[l - 1], pl.act[l],
                             (g.flags & SVAE_FLAG_RESID) != 0, (fuse_logits && l == g.L - 1) ? g.C : 0, p->out_w,
                             /*rows_ready=*/split_l0 && l == 1, st);
            continue;
        }
        DenseArgs a;
        a.in = pl.act[l - 1];
        a.wp = pl.wf[l - 1];
        a.out = pl.act[l];
        a.bias = p->hidden_b[l - 1];
        a.aux = nullptr;
        a.tiles = g.tiles;
        a.Hp = g.Hp;
        a.H = g.H;
        a.act = g.act;
        a.resid = (g.flags & SVAE_FLAG_RESID) ? 1 : 0;
        a.pose = pa; a.posebuf = nullptr; a.tab = nullptr; a.sgtile = nullptr; a.dfpart = nullptr;
        a.Mp = g.Mp; a.N = g.N; a.Timg = g.Timg;
        a.do_p = nullptr; a.out_w = p->out_w; a.C = g.C;
        a.lpart = pl.dfpart;  // free during the forward pass; nblk * C * Mp <= ntile * 2 * Mp floats
        const DenseBlocks used = launch_dense<false>(g, a, st, false, 0, (fuse_logits && l == g.L - 1) ? g.C : 0);
        if (l == g.L - 1) cfb = used;
    }
    if (fuse_logits) {
        Scope prof(K_OUT_FWD, st);
        const int nblk = g.ntile / (split ? split_nt_fwd(g) : cfb.nt);
        const long m_split = split ? (long)g.Mp : cfb.m_split;
        const int nblk_tail = g.ntile / cfb.nt_tail;
        if (bce_target) {
            // one block per (image, chunk of <= 1024 pixels); with several chunks per image (galaxy: 16 384 pixels) their
            // sums go to llpart and are added in chunk order by loglik_reduce_kernel (fixed order, no atomics)
            const int chunks = finish_chunks(g.N);
            hipLaunchKernelGGL(logits_finish_bce_kernel, dim3(g.B, chunks), dim3(g.N > 512 ? 1024 : (g.N > 256 ? 512 : 256)), 0, st,
                               pl.dfpart, p->out_b, bce_target, y, logits, chunks > 1 ? pl.llpart : loglik, dll_dy, row_geo(g), g.C,
                               nblk, (long)g.Mp, m_split, nblk_tail);
            if (chunks > 1)
                hipLaunchKernelGGL(loglik_reduce_kernel, dim3(blocks_for(g.B)), dim3(256), 0, st, pl.llpart, loglik, g.B, chunks);
        }
        else
            hipLaunchKernelGGL(logits_finish_kernel, dim3(blocks_for((long)g.B * g.N)), dim3(256), 0, st, pl.dfpart, p->out_b, y,
                               logits, row_geo(g), g.C, nblk, (g.flags & SVAE_FLAG_SOFTPLUS) ? 1 : 0, (long)g.Mp, m_split, nblk_tail);
        return launch_status("svae_decoder_forward");
    }
    switch (g.C) {
        case 1: launch_out_fwd<1>(g, pl.act[g.L - 1], p, y, logits, st); break;
        case 2: launch_out_fwd<2>(g, pl.act[g.L - 1], p, y, logits, st); break;
        case 3: launch_out_fwd<3>(g, pl.act[g.L - 1], p, y, logits, st); break;
        default: launch_out_fwd<4>(g, pl.act[g.L - 1], p, y, logits, st); break;
    }
    if (bce_target) {  // no fused finish for this geometry (C > 2 or no hidden GEMM): the plain per-image pass over y
        Scope prof(K_BCE, st);
        hipLaunchKernelGGL(bce_kernel, dim3(g.B), dim3(256), 0, st, y, bce_target, loglik, dll_dy, g.N * g.C);
    }
    return launch_status("svae_decoder_forward");
}
}  // namespace

extern "C" {

int svae_decoder_forward(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z, float* y,
                         float* logits, void* saved, void* ws, size_t ws_bytes, svae_stream_t stream) {
    return decoder_forward_impl(d, p, pose, z, y, logits, saved, ws, ws_bytes, stream, nullptr, nullptr, nullptr);
}

int svae_decoder_forward_bce(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z,
                             const float* target, float* y, float* logits, float* loglik, float* dll_dy, void* saved, void* ws,
                             size_t ws_bytes, svae_stream_t stream) {
    if (!target) return fail(SVAE_E_INVALID, "svae_decoder_forward_bce: target is null");
    return decoder_forward_impl(d, p, pose, z, y, logits, saved, ws, ws_bytes, stream, target, loglik, dll_dy);
}

int svae_decoder_backward(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z,
                          const float* logits, const float* dy, const float* dy_scale, const void* saved,
                          const svae_grads* grads, float* dz, const svae_pose_grads* pg, void* ws, size_t ws_bytes,
                          svae_stream_t stream) {
    int rc;
    if ((rc = check_desc(d)) || (rc = check_params(d, p)) || (rc = check_pose(pose))) return rc;
    if (!logits || !dy || !saved || !grads) return fail(SVAE_E_INVALID, "logits, dy, saved and grads are required");
    if (d->Zd > 0 && !z) return fail(SVAE_E_INVALID, "latent_dim > 0 but z is null");
    if (reinterpret_cast<uintptr_t>(saved) & 255) return fail(SVAE_E_WORKSPACE, "saved not 256-byte aligned");
    if (pg && (pg->dtheta || pg->ddx) && pose->coords)
        return fail(SVAE_E_INVALID, "dtheta/ddx requested but the pose was given as explicit coords");
    const Geo g = make_geo(*d);
    {
        const long rec = recorded_plan(saved);
        if (rec >= 0 && (unsigned)rec != plan_bits(g))
            return fail(SVAE_E_INVALID, "svae_decoder_backward: `saved` was written by a forward call planned differently (plan "
                        "bits %ld then, %u now): the GEMM mode (svae_gemm_mode_set / SVAE_GEMM) or SVAE_FUSE_OUT changed between "
                        "the forward call and its backward call", rec, plan_bits(g));
    }
    const Plan pl = make_plan(g, const_cast<void*>(saved), ws);
    if ((rc = check_ws(pl, ws, ws_bytes))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PoseArgs pa = pose_args(pose);
    const int resid = (g.flags & SVAE_FLAG_RESID) ? 1 : 0;
    // the packed weights, first-layer tables and poses were left in `saved` by the forward call (make_plan)
    // fp16x3: the data gradient of the last hidden layer runs on the f16 pipe (bounded act', contraction multiple of 64)
    const int fmode = fuse_out_mode();
    const bool split_bwd = split_active(g) && g.L >= 2 && fmode != 1;
    const bool r1 = rank1_out(g);   // the forward call packed this layer's data-gradient weights for it

    // d(loss)/d(logits) in padded row space (pad rows exactly zero)
    {
        Scope prof(K_DLOGITS, st);
        if (split_bwd && hipMemsetAsync(pl.amax, 0, sizeof(unsigned), st) != hipSuccess) return fail(SVAE_E_LAUNCH, "memset failed");
        hipLaunchKernelGGL(dlogits_kernel, dim3(blocks_for((long)g.C * g.Mp)), dim3(256), 0, st, logits, dy, dy_scale,
                           pl.do_p, g.B, g.N, g.Npad, g.C, (g.flags & SVAE_FLAG_SOFTPLUS) ? 1 : 0, (long)g.Mp,
                           split_bwd ? pl.amax : (unsigned*)nullptr);
        if (split_bwd)
            hipLaunchKernelGGL(split_scale_kernel, dim3(1), dim3(256), 0, st, pl.amax, p->out_w, g.C, g.H, pl.gscale);
    }

    // Output layer.  With at least one hidden layer and no residual, dh_{L-1} is never materialised: both
    // GEMMs of the last hidden layer form it from a_{L-1} on the fly (LASTW / LASTD) and the weight-gradient
    // kernel also produces dW_o, db_o.  Otherwise: one streaming pass (out_bwd_kernel).
    int cur = 0;
    // rank-1 form by default where it applies; the generic fused form is opt-in (slower on fp32 MFMA, saves a 419 MB plane)
    const bool fused_out = r1 || (fmode == 1 && g.L >= 2 && !resid && (grads->hidden_w[g.L - 2] || grads->hidden_b[g.L - 2]));
    // fp16x3 with both GEMMs of the last hidden layer on the f16 pipe: dh leaves out_bwd in their operand forms
    int ob_nparts = 0;
    static const bool ob_env = [] { const char* e = getenv("SVAE_SPLIT_OB"); return !(e && e[0] == '0'); }();
    const bool split_ob = ob_env && split_bwd && split_wgrad_on() && (grads->hidden_w[g.L - 2] || grads->hidden_b[g.L - 2]);
    if (split_ob) {
        const float* alast = pl.act[g.L - 1];
        float* dh32 = resid ? pl.dh[cur] : nullptr;  // only a residual epilogue reads dh itself
        const int nparts = g.act == SVAE_ACT_TANH ? launch_out_bwd_split_a<SVAE_ACT_TANH>(g, pl, alast, p, dh32, st)
                                                  : launch_out_bwd_split_a<SVAE_ACT_SIGMOID>(g, pl, alast, p, dh32, st);
        ob_nparts = nparts;
        Scope prof(K_SMALL_BWD, st);
        hipLaunchKernelGGL(out_bwd_reduce_kernel, dim3(g.C * ((g.Hp + 63) / 64) + 1), dim3(1024), 0, st, pl.wpart, pl.bpart,
                           grads->out_w, grads->out_b, g.C, g.H, g.Hp, nparts);
    } else if (!fused_out) {
        const float* alast = pl.act[g.L - 1];
        switch (g.act) {
            case SVAE_ACT_TANH: launch_out_bwd_a<SVAE_ACT_TANH>(g, pl, alast, p, pl.dh[cur], st); break;
            case SVAE_ACT_LEAKYRELU: launch_out_bwd_a<SVAE_ACT_LEAKYRELU>(g, pl, alast, p, pl.dh[cur], st); break;
            case SVAE_ACT_RELU: launch_out_bwd_a<SVAE_ACT_RELU>(g, pl, alast, p, pl.dh[cur], st); break;
            default: launch_out_bwd_a<SVAE_ACT_SIGMOID>(g, pl, alast, p, pl.dh[cur], st); break;
        }
        Scope prof(K_SMALL_BWD, st);
        hipLaunchKernelGGL(out_bwd_reduce_kernel, dim3(g.C * ((g.Hp + 63) / 64) + 1), dim3(1024), 0, st, pl.wpart, pl.bpart,
                           grads->out_w, grads->out_b, g.C, g.H, g.Hp, pl.ob_chunks * 2);
    }

    // hidden layers, last to first
    bool fused_first = false;
    bool tail_pending = false;
    WgradReduceArgs tail_r{};
    DenseBlocks fb{dense_nt_first(g.ntile), (long)g.Mp, 1};  // how the fp32 launch that ran the FIRST epilogue was blocked
    for (int l = g.L - 1; l >= 1; --l) {
        const bool last = fused_out && l == g.L - 1;
        // fp16x3 runs down the stack: every split data gradient leaves the scale of its result for the layer below
        // (SVAE_SPLIT_CHAIN=0: only the last hidden layer)
        const bool split_here = split_bwd && (l == g.L - 1 || split_chain_on());
        // the fused forms also get dW_o / db_o from this layer's weight-gradient kernel
        if (grads->hidden_w[l - 1] || grads->hidden_b[l - 1] || (last && (grads->out_w || grads->out_b))) {
            WgradArgs w;
            w.dh = last ? pl.act[l] : pl.dh[cur];
            w.aprev = pl.act[l - 1];
            w.slab = pl.slab;
            w.bslab = pl.bslab;
            w.noct = g.noct;
            w.vo = (g.N + 7) / 8; w.po = g.Npad / 8;
            w.Hp = g.Hp;
            w.nblk1 = pl.wg_nblk1;
            w.do_p = pl.do_p; w.out_w = p->out_w; w.wpart = pl.wpart; w.bpart = pl.bpart;
            w.Mp = g.Mp; w.H = g.H; w.act = g.act;
            if (split_here && split_wgrad_on())
                launch_split_wgrad(g, pl, pl.dh[cur], pl.act[l - 1],
                                   (l == 1 && g.L == 2 && split_l0_on()) ? pl.savedC : (const uint4*)nullptr,
                                   split_ob && l == g.L - 1, st);
            else {
                static const bool xcd_grid = [] { const char* e = getenv("SVAE_XCD_GRID"); return !(e && e[0] == '0'); }();
                launch_wgrad(w, pl.wg_S, xcd_grid, last ? g.C : 0, (last && r1) ? (g.act == SVAE_ACT_TANH ? 1 : 2) : 0, st);
            }
            Scope prof(K_WGRAD_REDUCE, st);
            const bool db_elsewhere = split_ob && l == g.L - 1 && split_wgrad_on();  // out_bwd_split summed dh's columns
            const WgradReduceArgs ra{pl.slab, pl.bslab, grads->hidden_w[l - 1],
                                     db_elsewhere ? (float*)nullptr : grads->hidden_b[l - 1], g.H, g.Hp, pl.wg_S,
                                     (last && r1) ? p->out_w : (const float*)nullptr};
            // the first hidden layer's reduction rides in the launch of the per-image sums behind this layer's data gradient
            // (backward_tail_kernel) when that launch exists: one latency-bound launch instead of two
            if (l == 1 && g.in_dim == 2 && !split_here && tail_merge_on()) {
                tail_r = ra;
                tail_pending = true;
            } else {
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((long)g.H * g.H + 63) / 64)), dim3(256), 0, st, ra);
            }
            if (db_elsewhere && grads->hidden_b[l - 1])
                hipLaunchKernelGGL(colsum_reduce_kernel, dim3(g.Hp / 32), dim3(256), 0, st, pl.hbpart, grads->hidden_b[l - 1],
                                   g.H, g.Hp, ob_nparts);
            if (last)
                hipLaunchKernelGGL(out_bwd_reduce_kernel, dim3(g.C * ((g.Hp + 63) / 64) + 1), dim3(1024), 0, st, pl.wpart, pl.bpart,
                                   grads->out_w, grads->out_b, g.C, g.H, g.Hp, pl.wg_S * 2);
        }
        DenseArgs a;
        a.in = last ? pl.act[l] : pl.dh[cur];
        a.wp = pl.wb[l - 1];
        a.out = pl.dh[cur ^ 1];
        a.bias = nullptr;
        a.aux = pl.act[l - 1];
        a.tiles = g.tiles;
        a.Hp = g.Hp;
        a.H = g.H;
        a.act = g.act;
        a.resid = resid;
        a.pose = pa; a.posebuf = pl.posebuf; a.tab = pl.tab; a.sgtile = pl.sgtile; a.dfpart = pl.dfpart;
        a.Mp = g.Mp; a.N = g.N; a.Timg = g.Timg;
        a.do_p = pl.do_p; a.out_w = p->out_w; a.C = g.C;
        fused_first = (l == 1) && g.in_dim == 2;
        if (split_here)
            launch_split_dgrad(g, pl, pl.dh[cur], p->hidden_w[l - 1], pl.act[l - 1], pl.dh[cur ^ 1], resid != 0, fused_first, pa,
                               split_ob && l == g.L - 1, st);
        else {
            const DenseBlocks used = launch_dense<true>(g, a, st, fused_first, last ? (r1 ? (g.act == SVAE_ACT_TANH ? 2 : 3) : 1) : 0);
            if (l == 1) fb = used;
        }
        cur ^= 1;
    }

    if (tail_pending && !fused_first) {   // (cannot happen today: the deferral and fused_first share their condition)
        tail_pending = false;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((long)g.H * g.H + 63) / 64)), dim3(256), 0, st, tail_r);
    }
    // coordinate layer
    const bool bil = (g.flags & SVAE_FLAG_BILINEAR) != 0;
    const bool want_coords = pg && (pg->dcoords || pg->dtheta || pg->ddx);
    float* dc = (pg && pg->dcoords) ? pg->dcoords : pl.dcoords;
    {
        Scope prof_l0(K_LAYER0_BWD, st);
        const bool split_first = split_bwd && (g.L == 2 || split_chain_on());   // which kernel ran the FIRST epilogue
        const bool want_dz = dz && g.Zd > 0;
        if (fused_first) {
            // dh0 was reduced inside the data-gradient GEMM's epilogue: only small fixed-order per-image sums remain
            const FirstLayerImageArgs fa{pl.sgtile, split_first ? 2 : 1, g.Timg, g.H, g.Hp, pl.sgimg, p->latent_w,
                                         bil ? p->bilinear_w : nullptr, want_dz ? dz : (float*)nullptr, g.Zd, g.in_dim, pl.dfpart,
                                         g.ntile / (split_first ? split_nt(g) : fb.nt), g.N, g.Npad, (long)g.Mp,
                                         want_coords ? dc : (float*)nullptr, pose->grid, pl.posebuf,
                                         pg ? pg->dtheta : (float*)nullptr, pg ? pg->ddx : (float*)nullptr,
                                         split_first ? (long)g.Mp : fb.m_split, g.ntile / fb.nt_tail};
            const unsigned nimg = (unsigned)g.B * (want_coords ? 2u : 1u);
            if (tail_pending) {
                tail_pending = false;
                hipLaunchKernelGGL(backward_tail_kernel, dim3(nimg + (unsigned)(((long)g.H * g.H + 63) / 64)), dim3(256), 0, st, fa,
                                   tail_r, nimg, (int)g.B);
            } else {
                hipLaunchKernelGGL(first_layer_image_kernel, dim3(g.B, want_coords ? 2 : 1), dim3(256), 0, st, fa);
            }
        } else {
            const float* dh0 = pl.dh[cur];
            hipLaunchKernelGGL(layer0_bwd_params_kernel, dim3(blocks_for(g.Hp * 2), g.B * pl.l0_chunks_per_image), dim3(256), 0,
                               st, pa, pl.posebuf, dh0, pl.sgpart, row_geo(g), pl.l0_oct_per_chunk, pl.l0_chunks_per_image);
            hipLaunchKernelGGL(sg_reduce_kernel, dim3(blocks_for((long)g.B * g.Hp * kSlots)), dim3(256), 0, st, pl.sgpart,
                               pl.sgimg, g.B, g.Hp, pl.l0_chunks_per_image);
            if (want_coords) {
                long nb = (g.noct + 3) / 4;
                if (nb > 4096) nb = 4096;
                hipLaunchKernelGGL(layer0_bwd_coords_kernel, dim3((unsigned)nb), dim3(256), 0, st, pa, pl.posebuf, dh0, pl.tab,
                                   dc, row_geo(g), g.noct);
            }
        }
        hipLaunchKernelGGL(layer0_param_grads_kernel, dim3((unsigned)(((long)g.H * kSlots + kPgCols - 1) / kPgCols)), dim3(256), 0, st,
                           pl.sgimg, z, grads->coord_w, grads->coord_b, g.Zd > 0 ? grads->latent_w : nullptr,
                           bil ? grads->bilinear_w : nullptr, g.B, g.H, g.Hp, g.Zd, g.in_dim);
        if (!fused_first) {
            if (want_dz)
                hipLaunchKernelGGL(dz_kernel, dim3(g.B), dim3(256), 0, st, pl.sgimg, p->latent_w, bil ? p->bilinear_w : nullptr,
                                   dz, g.H, g.Hp, g.Zd, g.in_dim);
            if (want_coords && (pg->dtheta || pg->ddx))
                hipLaunchKernelGGL(pose_bwd_kernel, dim3(g.B), dim3(256), 0, st, dc, pose->grid, pl.posebuf, pg->dtheta, pg->ddx,
                                   g.N);
        }
    }
    return launch_status("svae_decoder_backward");
}

int svae_bce_loglik(int32_t B, int32_t n, const float* y_hat, const float* target, float* loglik, float* dll_dy,
                    svae_stream_t stream) {
    if (B < 1 || n < 1 || !y_hat || !target || !loglik) return fail(SVAE_E_INVALID, "svae_bce_loglik: bad arguments");
    Scope prof(K_BCE, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(bce_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), y_hat, target, loglik, dll_dy, n);
    return launch_status("svae_bce_loglik");
}

size_t svae_gaussian_workspace_bytes(int32_t B, int32_t N) {
    if (B < 1 || N < 1) return 0;
    return (((size_t)B * N * sizeof(float) + 255) & ~size_t(255)) * 2;
}

int svae_gaussian_loglik(int32_t B, int32_t N, int32_t C, const float* y_params, const float* target,
                         const uint8_t* mask, const float* ctf, int32_t k, float* loglik, float* dll_dy, void* ws,
                         size_t ws_bytes, svae_stream_t stream) {
    if (B < 1 || N < 1 || !y_params || !target || !loglik) return fail(SVAE_E_INVALID, "svae_gaussian_loglik: bad arguments");
    if (C != 1 && C != 2) return fail(SVAE_E_INVALID, "Gaussian likelihood needs n_out 1 or 2 (got %d)", C);
    float* filt = nullptr;
    float* dflt = nullptr;
    if (ctf) {
        // the reference applies the filter to the variance without groups= and raises
        if (C != 1) return fail(SVAE_E_INVALID, "CTF with fit-noise has no defined result in the reference");
        const int n = (int)(sqrt((double)N) + 0.5);
        if (n * n != N) return fail(SVAE_E_INVALID, "CTF needs a square image (N = %d)", N);
        if (k < 1 || (k & 1) == 0) return fail(SVAE_E_INVALID, "CTF filter size must be odd (got %d)", k);
        const size_t half = ((size_t)B * N * sizeof(float) + 255) & ~size_t(255);
        if (!ws || ws_bytes < 2 * half) return fail(SVAE_E_WORKSPACE, "svae_gaussian_loglik: workspace too small");
        filt = static_cast<float*>(ws);
        dflt = reinterpret_cast<float*>(static_cast<char*>(ws) + half);
    }
    Scope prof(K_GAUSSIAN, static_cast<hipStream_t>(stream));
    if (ctf) {  // the correlation and its adjoint out of LDS when the padded image and the filter fit (they do up to ~100 x 100)
        const int n = (int)(sqrt((double)N) + 0.5);
        const CtfLds cg = CtfLds::make(n, k);
        const size_t lds = ((size_t)cg.W * cg.Wp + (size_t)k * cg.kp) * sizeof(float);
        static const bool lds_env = [] { const char* e = getenv("SVAE_CTF_LDS"); return !(e && e[0] == '0'); }();
        if (lds_env && lds <= 150 * 1024) {
            if (lds > 48 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(gaussian_ctf_lds_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return fail(SVAE_E_LAUNCH, "svae_gaussian_loglik: cannot reserve %zu bytes of LDS", lds);
            hipLaunchKernelGGL(gaussian_ctf_lds_kernel, dim3(B), dim3(kCtfThreads), lds, static_cast<hipStream_t>(stream),
                               y_params, target, mask, ctf, k, loglik, dll_dy, filt, dflt, N);
            return launch_status("svae_gaussian_loglik");
        }
    }
    hipLaunchKernelGGL(gaussian_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), y_params, target, mask, ctf,
                       k, loglik, dll_dy, filt, dflt, N, C);
    return launch_status("svae_gaussian_loglik");
}

static int latent_geo(const svae_latent_desc* d, LatentGeo* g) {
    if (!d || d->B < 1 || d->inf_dim < 1) return fail(SVAE_E_INVALID, "svae_latent: bad descriptor");
    if (d->inf_dim < (d->rotate ? 1 : 0) + (d->translate ? 2 : 0))
        return fail(SVAE_E_INVALID, "svae_latent: inf_dim %d too small for the requested pose", d->inf_dim);
    g->B = d->B; g->inf = d->inf_dim; g->rotate = d->rotate ? 1 : 0; g->translate = d->translate ? 1 : 0;
    g->mu_penalty = d->mu_penalty ? 1 : 0; g->dx_scale = d->dx_scale; g->z_scale = d->z_scale; g->theta_prior = d->theta_prior;
    return SVAE_OK;
}

int svae_latent_forward(const svae_latent_desc* d, const float* q_out, const float* r, float* theta, float* dx, float* zc,
                        float* kl, svae_stream_t stream) {
    LatentGeo g;
    int rc;
    if ((rc = latent_geo(d, &g))) return rc;
    const int zd = g.inf - g.rotate - 2 * g.translate;
    if (!q_out || !r || !kl || (g.rotate && !theta) || (g.translate && !dx) || (zd > 0 && !zc))
        return fail(SVAE_E_INVALID, "svae_latent_forward: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_LATENT, st);
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(blocks_for(g.B)), dim3(256), 0, st, q_out, r, theta, dx, zc, kl, g);
    return launch_status("svae_latent_forward");
}

int svae_latent_backward(const svae_latent_desc* d, const float* q_out, const float* r, const float* g_theta,
                         const float* g_dx, const float* g_zc, const float* g_kl, float* g_q_out, svae_stream_t stream) {
    LatentGeo g;
    int rc;
    if ((rc = latent_geo(d, &g))) return rc;
    if (!q_out || !r || !g_q_out) return fail(SVAE_E_INVALID, "svae_latent_backward: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_LATENT, st);
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(blocks_for((long)g.B * g.inf)), dim3(256), 0, st, q_out, r, g_theta, g_dx,
                       g_zc, g_kl, g_q_out, g);
    return launch_status("svae_latent_backward");
}

int svae_elbo_head_forward(const float* loglik, const float* kl, int32_t B, float* out3, svae_stream_t stream) {
    if (!loglik || !kl || !out3 || B < 1) return fail(SVAE_E_INVALID, "svae_elbo_head_forward: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_LATENT, st);
    hipLaunchKernelGGL(elbo_head_fwd_kernel, dim3(1), dim3(256), 0, st, loglik, kl, B, out3);
    return launch_status("svae_elbo_head_forward");
}

int svae_elbo_head_backward(const float* g_elbo, const float* g_logp, const float* g_kl, int32_t B, float* dloglik, float* dkl,
                            svae_stream_t stream) {
    if (!dloglik || !dkl || B < 1) return fail(SVAE_E_INVALID, "svae_elbo_head_backward: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_LATENT, st);
    hipLaunchKernelGGL(elbo_head_bwd_kernel, dim3(blocks_for(B)), dim3(256), 0, st, g_elbo, g_logp, g_kl, B, dloglik, dkl);
    return launch_status("svae_elbo_head_backward");
}

int svae_colsum(const float* x, int32_t rows, int32_t cols, float* out, svae_stream_t stream) {
    if (!x || !out || rows < 1 || cols < 1) return fail(SVAE_E_INVALID, "svae_colsum: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_SMALL_BWD, st);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, st, x, out, cols, cols, rows);
    return launch_status("svae_colsum");
}

int svae_linear_forward(const float* x, const float* weight, const float* bias, float* out, int32_t rows, int32_t in_features,
                        int32_t out_features, int32_t act, svae_stream_t stream) {
    if (!x || !weight || !out || rows < 1 || in_features < 1 || out_features < 1)
        return fail(SVAE_E_INVALID, "svae_linear_forward: bad arguments");
    if (act < SVAE_LINEAR_ACT_NONE || act > SVAE_ACT_SIGMOID) return fail(SVAE_E_INVALID, "svae_linear_forward: unknown activation %d", act);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_ENCODER, st);
    LinFwdArgs a{x, weight, bias, out, rows, in_features, out_features, act};
    const long tiles = (long)((rows + 15) / 16) * ((out_features + 15) / 16);   // one workgroup (4 waves) per 16 x 16 tile
    const bool vec = in_features % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(weight)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(enc_linear_fwd_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(enc_linear_fwd_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, st, a);
    return launch_status("svae_linear_forward");
}

int svae_linear_backward(const float* x, const float* weight, const float* out, const float* dout, int32_t rows,
                         int32_t in_features, int32_t out_features, int32_t act, float* dweight, float* dbias, float* dx,
                         svae_stream_t stream) {
    if (!x || !weight || !dout || rows < 1 || in_features < 1 || out_features < 1)
        return fail(SVAE_E_INVALID, "svae_linear_backward: bad arguments");
    if (act < SVAE_LINEAR_ACT_NONE || act > SVAE_ACT_SIGMOID) return fail(SVAE_E_INVALID, "svae_linear_backward: unknown activation %d", act);
    if (act != SVAE_LINEAR_ACT_NONE && !out) return fail(SVAE_E_INVALID, "svae_linear_backward: the layer's output is needed for act'");
    if (!dweight && !dbias && !dx) return SVAE_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_ENCODER, st);
    LinBwdArgs a{x, weight, out, dout, dweight, dbias, dx, rows, in_features, out_features, act, 0};
    const long tk = (in_features + 15) / 16;
    a.tiles_w = (dweight || dbias) ? (int)(((out_features + 15) / 16) * tk) : 0;
    const long groups = a.tiles_w + (dx ? (long)((rows + 15) / 16) * tk : 0);   // one workgroup per tile of dW or dx
    const bool vecx = out_features % 4 == 0 && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vecx) hipLaunchKernelGGL(enc_linear_bwd_kernel<true>, dim3((unsigned)groups), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(enc_linear_bwd_kernel<false>, dim3((unsigned)groups), dim3(256), 0, st, a);
    return launch_status("svae_linear_backward");
}

int svae_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, int64_t step, int32_t zero_grad, svae_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return fail(SVAE_E_INVALID, "svae_adam_step: bad arguments");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return fail(SVAE_E_INVALID, "svae_adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_ADAM, st);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, (long)n,
                       (float)(lr / bc1), (float)sqrt(bc2), beta1, beta2, eps, zero_grad ? 1 : 0);
    return launch_status("svae_adam_step");
}

int svae_rotate_bicubic(const float* y, float* y_rot, const double* matrix, const int32_t* quarter, int32_t B, int32_t rows,
                        int32_t cols, int32_t C, int32_t quantize_u8, svae_stream_t stream) {
    if (!y || !y_rot || !matrix || !quarter || y == y_rot) return fail(SVAE_E_INVALID, "svae_rotate_bicubic: bad pointers");
    if (B < 1 || rows < 1 || cols < 1 || C < 1 || (long)B * rows * cols * C > (1L << 40))
        return fail(SVAE_E_INVALID, "svae_rotate_bicubic: bad sizes B=%d rows=%d cols=%d C=%d", B, rows, cols, C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Scope prof(K_AUGMENT, st);
    const RotGeo g{B, rows, cols, C};
    const long total = (long)B * rows * cols * C;
    if (quantize_u8)
        hipLaunchKernelGGL(rotate_bicubic_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, st, y, y_rot, matrix, quarter, g);
    else
        hipLaunchKernelGGL(rotate_bicubic_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, st, y, y_rot, matrix, quarter, g);
    return launch_status("svae_rotate_bicubic");
}

namespace {
const size_t CTF_LDS_MAX = 160 * 1024;
const int CTF_SCRATCH_GROUPS = 512;  // workgroups of the scratch form (two per CU); each strides over the particles
size_t ctf_lds_bytes(int n, int m) { return ((size_t)n * m * 3 + 2 * ((size_t)n + m)) * sizeof(double); }
}  // namespace

size_t svae_ctf_filter_workspace_bytes(int32_t count, int32_t n, int32_t m) {
    if (count < 1 || n < 1 || m < 1) return 0;
    if (ctf_lds_bytes(n, m) <= CTF_LDS_MAX) return 0;  // the whole transform fits the LDS of one CU
    const size_t groups = (size_t)(count < CTF_SCRATCH_GROUPS ? count : CTF_SCRATCH_GROUPS);
    return groups * (size_t)n * m * 3 * sizeof(double);
}

int svae_ctf_filter(const double* params, float* filters, int32_t count, int32_t n, int32_t m, double scale, void* ws,
                    size_t ws_bytes, svae_stream_t stream) {
    if (!params || !filters || count < 1 || n < 1 || m < 1) return fail(SVAE_E_INVALID, "svae_ctf_filter: bad arguments");
    // the scratch form keeps only the 2 (n + m) twiddle factors in LDS: 16 (n + m) bytes must fit one CU's 160 KiB
    const size_t tw = 2 * ((size_t)n + m) * sizeof(double);
    if (tw > CTF_LDS_MAX)
        return fail(SVAE_E_INVALID, "svae_ctf_filter: %d x %d filters are not supported (n + m must not exceed %zu)", n, m,
                    CTF_LDS_MAX / (2 * sizeof(double)));
    if (!(scale > 0.0)) return fail(SVAE_E_INVALID, "svae_ctf_filter: scale must be positive");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = ctf_lds_bytes(n, m);
    Scope prof(K_AUGMENT, st);
    if (lds <= CTF_LDS_MAX) {
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(ctf_filter_kernel<false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(SVAE_E_LAUNCH, "svae_ctf_filter: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL(ctf_filter_kernel<false>, dim3(count), dim3(256), lds, st, params, filters, count, n, m, scale,
                           static_cast<double*>(nullptr));
    } else {
        const size_t need = svae_ctf_filter_workspace_bytes(count, n, m);
        if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255))
            return fail(SVAE_E_WORKSPACE, "svae_ctf_filter: %d x %d filters need %zu bytes of 256-byte aligned scratch (got %zu)",
                        n, m, need, ws_bytes);
        const int groups = count < CTF_SCRATCH_GROUPS ? count : CTF_SCRATCH_GROUPS;
        if (tw > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(ctf_filter_kernel<true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tw) != hipSuccess)
            return fail(SVAE_E_LAUNCH, "svae_ctf_filter: cannot reserve %zu bytes of LDS", tw);
        hipLaunchKernelGGL(ctf_filter_kernel<true>, dim3(groups), dim3(256), tw, st, params, filters, count, n, m, scale,
                           static_cast<double*>(ws));
    }
    return launch_status("svae_ctf_filter");
}

int svae_gemm_mode_set(int mode) {
    if (mode != SVAE_GEMM_FP32 && mode != SVAE_GEMM_FP16X3) return fail(SVAE_E_INVALID, "svae_gemm_mode_set: unknown mode %d", mode);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_gemm_mode = mode;
    return SVAE_OK;
}

int svae_gemm_mode_get(void) { return split_mode() ? SVAE_GEMM_FP16X3 : SVAE_GEMM_FP32; }

int svae_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_level = on < 0 ? 0 : (on > 2 ? 2 : on);
    return SVAE_OK;
}

int svae_profile_read(double* ms_total, int64_t* launches) {
    if (!ms_total || !launches) return fail(SVAE_E_INVALID, "svae_profile_read: null output");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int k = 0; k < SVAE_PROF_KINDS; ++k) {
        ms_total[k] = 0.0;
        launches[k] = 0;
    }
    for (const ProfRec& r : g_prof_used) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess)
            return fail(SVAE_E_LAUNCH, "svae_profile_read: event query failed");
        ms_total[r.kind] += ms;
        launches[r.kind] += 1;
        g_prof_free.push_back(r);
    }
    g_prof_used.clear();
    return SVAE_OK;
}

const char* svae_profile_kind_name(int kind) { return (kind >= 0 && kind < SVAE_PROF_KINDS) ? kKindNames[kind] : ""; }

int svae_path_counts(int64_t* counts, int reset) {
    if (!counts) return fail(SVAE_E_INVALID, "svae_path_counts: null output");
    for (int i = 0; i < SVAE_PATH_KINDS; ++i)
        counts[i] = reset ? (int64_t)g_path[i].exchange(0, std::memory_order_relaxed) : (int64_t)g_path[i].load(std::memory_order_relaxed);
    return SVAE_OK;
}
const char* svae_path_name(int path) { return (path >= 0 && path < SVAE_PATH_KINDS) ? kPathNames[path] : ""; }

}  // extern "C"
