// Tile <-> patch exchange over RCCL, inside the library (include/scythe_hip.h "exchange over RCCL").
//
// The reference moves the 3-coefficient halo through a RemoteChannel chain and sums the tiles' B coefficients in a
// SharedArray on the master (src/semiimplicit.jl:203-219, 320-329, 272-282), then every worker solves the whole patch
// (:285).  Here one process owns one GPU and one tile; the exchange runs on the handle's stream with ncclSend / ncclRecv /
// ncclAllGather, so a host in any language (the Julia glue of INTEGRATION.md) needs nothing but ccall - no torch.
// librccl is bound at first use with dlopen: a single-GPU user never loads it.
#include "sx_internal.hpp"
#include <dlfcn.h>
#include <cstdlib>
#include <algorithm>
#include <cstring>

namespace sx {

// the parts of rccl.h this file uses (ABI of RCCL 2.x: /opt/rocm/include/rccl/rccl.h:40-43, 187, 220, 260, 339, 467, 678-722, 923-933)
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
enum { NCCL_DOUBLE = 8 };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};

static Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char *forced = getenv("SX_RCCL_LIB")) {
        // an explicit choice is final (also over a copy PyTorch has already mapped): another RCCL build, or the stand-in
        // transport of the tests (tests/fake_rccl.cpp)
        r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    } else {
        // a copy that is already mapped (e.g. the one PyTorch ships) wins: one RCCL per process
        for (const char *n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) { set_error(std::string("librccl not found: ") + dlerror()); return nullptr; }
#define BIND(field, sym)                                                                  \
    *(void **)(&r.field) = dlsym(r.lib, sym);                                             \
    if (!r.field) { set_error(std::string("librccl lacks ") + sym); r.lib = nullptr; return nullptr; }
    BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
    BIND(GetErrorString, "ncclGetErrorString") BIND(Send, "ncclSend") BIND(Recv, "ncclRecv") BIND(AllGather, "ncclAllGather")
    BIND(GroupStart, "ncclGroupStart") BIND(GroupEnd, "ncclGroupEnd")
#undef BIND
    return &r;
}

#define NCCLOK(call)                                                                       \
    do {                                                                                   \
        const int rc_ = (call);                                                            \
        if (rc_ != 0) { set_error(std::string(#call) + ": " + R->GetErrorString(rc_)); return 1; }   \
    } while (0)

// ncclGroupStart / ncclGroupEnd as a scope: an early return between the two (a failed ncclSend / ncclRecv) must still close
// the thread's group, or every later RCCL call of this thread - torch's included - is queued and never launched
struct GroupScope {
    Rccl *R;
    bool open = false;
    explicit GroupScope(Rccl *r) : R(r) {}
    int start() { const int rc = R->GroupStart(); open = (rc == 0); return rc; }
    int end() { open = false; return R->GroupEnd(); }
    ~GroupScope() { if (open) R->GroupEnd(); }
};

struct CommState {
    NcclComm comm = nullptr;
    bool owned = false;            // created by sx_comm_init (destroyed with the handle) vs attached by the host
    int n = 0, me = 0, mode = -1;  // mode 0: transposed solve (all-to-all), 1: halo + all-gather + redundant solve,
                                   // 2: interface-only solve (tile-local solves + all-to-all of 10 rows, sx_iface.hip)
    std::vector<int> cell0, ncells;
    // a2a: tile side [dest d][row][cw[d]], owner side [tile t][row][cw[me]]
    std::vector<int64_t> tile_off, tile_cnt, own_off, own_cnt;
    double *tile_buf = nullptr, *tile_buf2 = nullptr, *own_in = nullptr, *own_out = nullptr;
    // gather: [tile][max_rows][C] + the 3 received halo rows
    double *gbuf = nullptr, *halo = nullptr;
    int64_t max_rows = 0;
    std::vector<void *> bufs;      // every device buffer of this state (also listed in sx_handle::allocs until released)
};

static bool dev_alloc(sx_handle *h, double **p, int64_t n) {
    void *d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(n, 1);
    if (hipMalloc(&d, bytes) != hipSuccess || hipMemset(d, 0, bytes) != hipSuccess) { set_error("hipMalloc failed (exchange buffers)"); return false; }
    h->allocs.push_back(d);
    h->dev_bytes += bytes;
    if (h->comm_state) ((CommState *)h->comm_state)->bufs.push_back(d);
    *p = (double *)d;
    return true;
}

void comm_release(sx_handle *h) {
    CommState *c = (CommState *)h->comm_state;
    if (!c) return;
    if (c->owned && c->comm) { Rccl *R = rccl(); if (R) R->CommDestroy(c->comm); }
    // the exchange buffers go with the state (a repeated sx_comm_init must not pile them up until sx_destroy); the kernels
    // may still be reading them, and the bindings into them must not outlive them
    if (!c->bufs.empty()) {
        hipStreamSynchronize(h->stream);
        sx_bind_tile_b(h, nullptr);
        sx_bind_patch_b(h, nullptr, nullptr);
        for (void *b : c->bufs) {
            auto it = std::find(h->allocs.begin(), h->allocs.end(), b);
            if (it != h->allocs.end()) h->allocs.erase(it);
            hipFree(b);
        }
    }
    delete c;
    h->comm_state = nullptr;
}

}  // namespace sx

using namespace sx;

extern "C" {

int sx_comm_unique_id(char *out128) {
    clear_error();
    Rccl *R = rccl();
    if (!R || !out128) return 1;
    NcclUniqueId id;
    NCCLOK(R->GetUniqueId(&id));
    std::memcpy(out128, id.internal, 128);
    return 0;
}

static int configure(sx_handle *h, CommState *c, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells, int32_t mode) {
    if (mode < 0 || mode > 2) { set_error("exchange mode must be 0 (all-to-all), 1 (gather) or 2 (interface-only)"); return 1; }
    if (mode == 2 && n == 1) mode = 0;          // one tile has no interfaces: the plain patch solve
    // both protocols index patch rows through this table (mode 1 writes a b_rDim-sized offset vector from it, and a tile of
    // fewer than 3 cells would make the halo rows it sends overlap the rows that receive the halo add)
    if (!tile_table_ok(h, n, me, cell0, ncells)) return 1;
    c->n = n; c->me = me; c->mode = mode;
    c->cell0.assign(cell0, cell0 + n);
    c->ncells.assign(ncells, ncells + n);
    if (mode == 0 || mode == 2) {
        // both all-to-all protocols: tile side [dest d][rows of mine][cw[d]], owner side [tile t][rows of t][cw[me]];
        // mode 0 moves all ncells + 3 rows of a tile, mode 2 its 10 interface rows
        std::vector<int64_t> cs(n + 1);
        if (mode == 0 ? (sx_a2a_configure(h, n, me, cell0, ncells) || sx_a2a_col_starts(h, cs.data()))
                      : (sx_iface_configure(h, n, me, cell0, ncells) || sx_iface_col_starts(h, cs.data()))) return 1;
        c->tile_off.assign(n, 0); c->tile_cnt.assign(n, 0); c->own_off.assign(n, 0); c->own_cnt.assign(n, 0);
        int64_t to = 0, oo = 0;
        for (int d = 0; d < n; d++) {
            const int64_t rows_me = mode == 0 ? ncells[me] + 3 : 10, rows_d = mode == 0 ? ncells[d] + 3 : 10;
            c->tile_off[d] = to; c->tile_cnt[d] = rows_me * (cs[d + 1] - cs[d]); to += c->tile_cnt[d];
            c->own_off[d] = oo; c->own_cnt[d] = rows_d * (cs[me + 1] - cs[me]); oo += c->own_cnt[d];
        }
        if (!dev_alloc(h, &c->tile_buf, to) || !dev_alloc(h, &c->tile_buf2, to) || !dev_alloc(h, &c->own_in, oo) || !dev_alloc(h, &c->own_out, oo)) return 1;
    } else {
        c->max_rows = 0;
        for (int t = 0; t < n; t++) c->max_rows = std::max<int64_t>(c->max_rows, ncells[t] + 3);
        if (!dev_alloc(h, &c->gbuf, (int64_t)n * c->max_rows * h->C) || !dev_alloc(h, &c->halo, 3 * h->C)) return 1;
        std::vector<int64_t> ro(h->b_rDim, 0);
        for (int t = 0; t < n; t++) {
            const int owned = ncells[t] + (t == n - 1 ? 3 : 0);
            for (int j = 0; j < owned; j++) ro[cell0[t] + j] = ((int64_t)t * c->max_rows + j) * h->C;
        }
        if (sx_bind_tile_b(h, c->gbuf + (int64_t)me * c->max_rows * h->C)) return 1;
        if (sx_bind_patch_b(h, c->gbuf, ro.data())) return 1;
    }
    return 0;
}

// Everything of the set-up that can fail on ONE rank alone - binding librccl, validating the tile table, the exchange buffers -
// happens here, before the collective ncclCommInitRank: a host can run this on every rank, agree on the outcome (any
// side channel will do) and only then enter sx_comm_init, which all ranks or none must enter.
static int prepare(sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells, int32_t mode) {
    if (!rccl()) return 1;
    comm_release(h);
    CommState *c = new CommState();
    h->comm_state = c;
    if (configure(h, c, n, me, cell0, ncells, mode)) {
        const std::string keep = sx_last_error();     // comm_release re-binds the B arrays, which clears the message
        comm_release(h);
        set_error(keep);
        return 1;
    }
    return 0;
}

static bool prepared_as(const sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells, int32_t mode) {
    const CommState *c = (const CommState *)h->comm_state;
    if (!c || c->comm || c->n != n || c->me != me || c->mode != mode) return false;
    for (int t = 0; t < n; t++)
        if (c->cell0[t] != cell0[t] || c->ncells[t] != ncells[t]) return false;
    return true;
}

int sx_comm_prepare(sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells, int32_t mode) {
    clear_error();
    if (!h || !cell0 || !ncells || n < 1 || me < 0 || me >= n) { set_error("sx_comm_prepare: invalid argument"); return 1; }
    return prepare(h, n, me, cell0, ncells, mode);
}

int sx_comm_init(sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells, int32_t mode, const char *id128) {
    clear_error();
    if (!h || !cell0 || !ncells || !id128 || n < 1 || me < 0 || me >= n) { set_error("sx_comm_init: invalid argument"); return 1; }
    if (!prepared_as(h, n, me, cell0, ncells, mode) && prepare(h, n, me, cell0, ncells, mode)) return 1;
    Rccl *R = rccl();
    CommState *c = (CommState *)h->comm_state;
    NcclUniqueId id;
    std::memcpy(id.internal, id128, 128);
    const int rc = R->CommInitRank(&c->comm, n, id, me);      // collective over the n tiles; uses the calling thread's current device
    if (rc != 0) {
        const std::string msg = std::string("ncclCommInitRank: ") + R->GetErrorString(rc);
        c->comm = nullptr;
        comm_release(h);                                      // never leave a half-configured state behind
        set_error(msg);
        return 1;
    }
    c->owned = true;
    return 0;
}

int sx_comm_attach(sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells, int32_t mode, void *nccl_comm) {
    clear_error();
    if (!h || !cell0 || !ncells || !nccl_comm || n < 1 || me < 0 || me >= n) { set_error("sx_comm_attach: invalid argument"); return 1; }
    if (!prepared_as(h, n, me, cell0, ncells, mode) && prepare(h, n, me, cell0, ncells, mode)) return 1;
    CommState *c = (CommState *)h->comm_state;
    c->comm = nccl_comm;
    c->owned = false;
    return 0;
}

// Loopback transport: all n tiles live in this process on one GPU.  The same buffer geometry and offsets as the RCCL path,
// with every (send, recv) pair replaced by a device-to-device copy on the receiving tile's stream - so the offset tables of
// sx_exchange are exercised by the single-GPU test suite for n = 2, 3, 4 tiles (RCCL itself refuses two ranks on one device).
int sx_comm_init_local(sx_handle **hs, int32_t n, const int32_t *cell0, const int32_t *ncells, int32_t mode) {
    clear_error();
    if (!hs || !cell0 || !ncells || n < 1) { set_error("sx_comm_init_local: invalid argument"); return 1; }
    for (int t = 0; t < n; t++) {
        if (!hs[t]) { set_error("sx_comm_init_local: null handle"); return 1; }
        comm_release(hs[t]);
        CommState *c = new CommState();
        hs[t]->comm_state = c;
        if (configure(hs[t], c, n, t, cell0, ncells, mode)) {
            const std::string keep = sx_last_error();
            for (int u = 0; u <= t; u++) comm_release(hs[u]);
            set_error(keep);
            return 1;
        }
    }
    return 0;
}

int sx_exchange_local(sx_handle **hs, int32_t n) {
    clear_error();
    if (!hs || n < 1) { set_error("sx_exchange_local: invalid argument"); return 1; }
    std::vector<CommState *> cs(n);
    for (int t = 0; t < n; t++) {
        if (!hs[t] || !hs[t]->comm_state) { set_error("sx_exchange_local: call sx_comm_init_local first"); return 1; }
        cs[t] = (CommState *)hs[t]->comm_state;
        if (cs[t]->n != n || cs[t]->me != t) { set_error("sx_exchange_local: handles are not tiles 0..n-1 of one patch"); return 1; }
    }
    auto copy = [&](double *dst, const double *src, int64_t cnt, hipStream_t s) {
        return cnt == 0 || hipMemcpyAsync(dst, src, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, s) == hipSuccess;
    };
    // all tiles share one device; run everything on tile 0's stream so that the copies are ordered with the kernels.  A tile
    // bound to another stream may still have work queued there: tile 0's stream first waits for it, and that stream waits
    // for the exchange afterwards.
    hipStream_t s0 = hs[0]->stream;
    std::vector<hipStream_t> keep(n);
    bool foreign = false;
    for (int t = 0; t < n; t++) { keep[t] = hs[t]->stream; foreign |= keep[t] != s0; }
    hipEvent_t ev = nullptr;
    if (foreign) {
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return 1; }
        for (int t = 1; t < n; t++)
            if (keep[t] != s0 && (hipEventRecord(ev, keep[t]) != hipSuccess || hipStreamWaitEvent(s0, ev, 0) != hipSuccess)) {
                hipEventDestroy(ev);
                set_error("stream ordering of sx_exchange_local failed");
                return 1;
            }
    }
    for (int t = 0; t < n; t++) hs[t]->stream = s0;
    int rc = 0;
    if (cs[0]->mode == 0 || cs[0]->mode == 2) {
        const bool ifc = cs[0]->mode == 2;
        for (int t = 0; t < n && !rc; t++) rc = ifc ? sx_iface_local(hs[t], cs[t]->tile_buf) : sx_a2a_pack_b(hs[t], cs[t]->tile_buf);
        for (int sdr = 0; sdr < n && !rc; sdr++)          // sender sdr -> receiver d: what ncclSend(sdr, d) / ncclRecv(d, sdr) move
            for (int d = 0; d < n; d++) {
                if (cs[sdr]->tile_cnt[d] != cs[d]->own_cnt[sdr]) { set_error("exchange geometry mismatch"); rc = 1; break; }
                if (!copy(cs[d]->own_in + cs[d]->own_off[sdr], cs[sdr]->tile_buf + cs[sdr]->tile_off[d], cs[sdr]->tile_cnt[d], s0)) { set_error("hipMemcpyAsync failed"); rc = 1; break; }
            }
        for (int t = 0; t < n && !rc; t++)
            rc = ifc ? sx_iface_reduce(hs[t], cs[t]->own_in, cs[t]->own_out) : sx_a2a_solve(hs[t], cs[t]->own_in, cs[t]->own_out);
        for (int o = 0; o < n && !rc; o++)                // owner o -> tile t
            for (int t = 0; t < n; t++)
                if (!copy(cs[t]->tile_buf2 + cs[t]->tile_off[o], cs[o]->own_out + cs[o]->own_off[t], cs[o]->own_cnt[t], s0)) { set_error("hipMemcpyAsync failed"); rc = 1; break; }
        for (int t = 0; t < n && !rc; t++) rc = ifc ? sx_iface_apply(hs[t], cs[t]->tile_buf2) : sx_a2a_unpack_a(hs[t], cs[t]->tile_buf2);
    } else {
        const int64_t C = hs[0]->C;
        for (int t = 1; t < n && !rc; t++) {              // halo rows t-1 -> t, then the halo add on t
            const double *src = cs[t - 1]->gbuf + ((int64_t)(t - 1) * cs[t - 1]->max_rows + cs[t - 1]->ncells[t - 1]) * C;
            if (!copy(cs[t]->halo, src, 3 * C, s0)) { set_error("hipMemcpyAsync failed"); rc = 1; break; }
            rc = sx_halo_add(hs[t], cs[t]->halo);
        }
        for (int t = 0; t < n && !rc; t++)                // all-gather: every tile's block into every other tile's buffer
            for (int o = 0; o < n; o++) {
                if (o == t) continue;
                const int64_t off = (int64_t)o * cs[o]->max_rows * C;
                if (!copy(cs[t]->gbuf + off, cs[o]->gbuf + off, cs[o]->max_rows * C, s0)) { set_error("hipMemcpyAsync failed"); rc = 1; break; }
            }
        for (int t = 0; t < n && !rc; t++) rc = sx_spline_transform(hs[t]);
    }
    for (int t = 0; t < n; t++) hs[t]->stream = keep[t];
    if (foreign) {
        bool ok = hipEventRecord(ev, s0) == hipSuccess;
        for (int t = 1; t < n && ok; t++)
            if (keep[t] != s0) ok = hipStreamWaitEvent(keep[t], ev, 0) == hipSuccess;
        hipEventDestroy(ev);
        if (!ok && !rc) { set_error("stream ordering of sx_exchange_local failed"); rc = 1; }
    }
    return rc;
}

int sx_exchange(sx_handle *h) {
    clear_error();
    if (!h || !h->comm_state) { set_error("sx_exchange: call sx_comm_init / sx_comm_attach first"); return 1; }
    CommState *c = (CommState *)h->comm_state;
    Rccl *R = rccl();
    if (!R) return 1;
    if (!c->comm) { set_error("sx_exchange: no communicator (sx_comm_prepare without sx_comm_init / sx_comm_attach, or tiles set up for sx_exchange_local)"); return 1; }
    const int n = c->n, me = c->me;
    hipStream_t s = h->stream;
    GroupScope grp(R);
    if (c->mode == 0 || c->mode == 2) {
        // transposed solve: B rows -> owners of the column ranges, solve my columns for the whole patch, A rows back;
        // interface-only solve: the same two all-to-alls with 10 rows per tile around the reduced system (sx_iface.hip)
        const bool ifc = c->mode == 2;
        if (ifc ? sx_iface_local(h, c->tile_buf) : sx_a2a_pack_b(h, c->tile_buf)) return 1;
        NCCLOK(grp.start());
        // a rank that owns no column (more ranks than columns to share out) exchanges nothing: its counts are zero on BOTH
        // sides of the pair (rows x columns of the owner), so sender and receiver skip the same messages
        for (int d = 0; d < n; d++) {
            if (c->tile_cnt[d]) NCCLOK(R->Send(c->tile_buf + c->tile_off[d], (size_t)c->tile_cnt[d], NCCL_DOUBLE, d, c->comm, s));
            if (c->own_cnt[d]) NCCLOK(R->Recv(c->own_in + c->own_off[d], (size_t)c->own_cnt[d], NCCL_DOUBLE, d, c->comm, s));
        }
        NCCLOK(grp.end());
        if (ifc ? sx_iface_reduce(h, c->own_in, c->own_out) : sx_a2a_solve(h, c->own_in, c->own_out)) return 1;
        NCCLOK(grp.start());
        for (int t = 0; t < n; t++) {
            if (c->own_cnt[t]) NCCLOK(R->Send(c->own_out + c->own_off[t], (size_t)c->own_cnt[t], NCCL_DOUBLE, t, c->comm, s));
            if (c->tile_cnt[t]) NCCLOK(R->Recv(c->tile_buf2 + c->tile_off[t], (size_t)c->tile_cnt[t], NCCL_DOUBLE, t, c->comm, s));
        }
        NCCLOK(grp.end());
        return ifc ? sx_iface_apply(h, c->tile_buf2) : sx_a2a_unpack_a(h, c->tile_buf2);
    }
    // the reference's protocol: halo rows tile -> tile + 1 (:320-329), owned rows to everybody (:272-282), redundant solve (:285)
    double *mine = c->gbuf + (int64_t)me * c->max_rows * h->C;
    if (n > 1) {
        NCCLOK(grp.start());
        if (me < n - 1) NCCLOK(R->Send(mine + (int64_t)c->ncells[me] * h->C, (size_t)(3 * h->C), NCCL_DOUBLE, me + 1, c->comm, s));
        if (me > 0) NCCLOK(R->Recv(c->halo, (size_t)(3 * h->C), NCCL_DOUBLE, me - 1, c->comm, s));
        NCCLOK(grp.end());
        if (me > 0 && sx_halo_add(h, c->halo)) return 1;
    }
    NCCLOK(R->AllGather(mine, c->gbuf, (size_t)(c->max_rows * h->C), NCCL_DOUBLE, c->comm, s));     // in place: my block is my slot
    return sx_spline_transform(h);
}

}  // extern "C"
