// One streamed tvdn_run on one device (include/tvdn.h, tvdn_run_args.stream_rows / stream_k): the state of the run and its
// named steps.  Round 4 had all of this as locals and lambdas of one 1270-line function; round 5 gave them names:
//   set_up()    tvdn_stream.hip        what fits where; host state page-locked by a helper thread under the first pass; rings,
//                                      boxes and resident rows carved from one kept device block; the resident rows' data term staged
//   pass()      tvdn_stream_pass.hip   one DRAINED pass of kk levels (periodic cubes, slabs of a device list or of a multi-process run)
//   chain()     tvdn_stream_chain.hip  Jia-Zhao on one device: one or several passes stacked into one running row index
//   schedule()  tvdn_stream.hip        which passes, in which form; stopping rule; progress
//   finish()    tvdn_stream.hip        results home, sums, stats, teardown in an order that does not stall
// The members keep the names the locals had, so that the bodies read as before.
#pragma once

#include "tvdn_stream_parts.hpp"

namespace tvdn {

struct PassDesc {
    int it0 = 0, kk = 0;         // first iteration slot, levels
    std::vector<int> modes;      // TVDN_ITER_* per level
    std::vector<double> tk, tkp; // momentum ratio of the level / of the level before it
    int n_in_state = 1, n_out_state = 1;
    bool first = false;          // starts from recon = data term and zero accumulators: uploads the data term only
    bool last = false;           // the run ends with this pass (no stopping rule): resident rows send their result straight home
};

struct StreamRun {
    // ---- what was asked ----------------------------------------------------------------------------------------------------
    const tvdn_run_args *a;
    int64_t R, K, res_req;
    const SlabShare *sh;
    StreamRun(const tvdn_run_args *a_, int64_t R_, int64_t K_, int64_t res_req_, const SlabShare *sh_) : a(a_), R(R_), K(K_), res_req(res_req_), sh(sh_) {}
    ~StreamRun()
    {
        if (stager.joinable()) stager.join();
        if (pinner.joinable()) pinner.join();
    }
    StreamRun(const StreamRun &) = delete;
    StreamRun &operator=(const StreamRun &) = delete;

    // ---- the cube and the schedule's geometry (set_up) -----------------------------------------------------------------------
    std::chrono::steady_clock::time_point t_start;
    int nd = 0, n_total = 0, n_state = 1, device = 0, n_pass_plan = 1;
    size_t item = 4, plane = 1, row_bytes = 0, cube_bytes = 0;
    int64_t N0 = 0;
    bool fista = false, want_mse = false, periodic = false, aliased = false;
    int64_t KX = 0, NV = 0, G0 = 0, G1 = 0, own0 = 0, own1 = 0;  // virtual rows: the cube between KX wrapped / neighbours' rows
    bool art_lo = false, art_hi = false;                        // faces that are not the cube's own
    int64_t cap = 0, ocap = 0;                                  // ring capacities (rows)
    int n_in = 0, n_out = 0, n_store = 0;
    size_t ring_b = 0, oring_b = 0, box_b = 0, plane_b = 0, obox_b = 0, dev_bytes_max = 0;
    RowMap rm;                                                  // which rows keep their state in HBM between passes
    int64_t RES = 0, HR = 0;                                    // rows in HBM / rows on the host
    bool exact_wrap = false;
    // ---- host state ------------------------------------------------------------------------------------------------------------
    HostArr orig_h, recon_h, ref_h, recon2_h;
    StateBlocks sb[2];
    bool two_sets = false;
    int n_sets = 1;
    Flag orig_ready, recon_ready, recon2_ready, staged_done;
    std::atomic<int64_t> staged_upto{0};  // resident rows g < staged_upto have their data term in the store
    size_t host_bytes = 0;
    // ---- device --------------------------------------------------------------------------------------------------------------
    CtxHolder ctx;
    Streams st;
    size_t ring_bytes = 0, store_b = 0, dev_bytes = 0;
    struct KeptBlock {  // the one big device block: kept for the next run when this one ends (tvdn_run.hip state_acquire / state_release)
        void *p = nullptr;
        size_t bytes = 0;
        int device = 0;
        void release()
        {
            if (p) state_release(p, bytes, device);
            p = nullptr;
        }
        ~KeptBlock() { release(); }
    } mem;
    DevMem sums_d, mse_d;
    double t_before_block = 0.0, t_block = 0.0;
    bool block_reused = false;
    int block_kind = 0;
    char *cursor = nullptr;
    std::vector<Ring> Rw, Aw;  // recon rings by level; accumulator rings [level + 1][axis]
    Ring Ow, Fw;               // data term, reference
    char *inbox[2][12], *outbox[2][12];
    char *zero_plane = nullptr;
    std::vector<char *> row0, row0b;  // row 0 of every level, kept for the top face (second set: two chained passes at a seam)
    char *row0b_base = nullptr;
    std::vector<char *> store;        // resident rows, packed: 0 data term, 1 recon, 2 + q * n_state + s state
    int discard = 0;
    Events evs;
    hipEvent_t in_ready[2], in_free[2], out_ready[2], out_free[2];
    bool in_free_set[2] = {false, false}, out_free_set[2] = {false, false};
    std::thread pinner, stager;  // helper threads: page-lock the host state / stage the resident rows' data term
    int h_old = 0;               // which set holds the current state (two sets: periodic runs, slabs)
    bool local_rows = false;
    tvdn_iter_args it;
    int64_t one_row[4];
    PinnedBuf row0_host;  // exact wrap across processes: row 0 of every level on its way from the first slab to the others
    // ---- progress of the run ---------------------------------------------------------------------------------------------------
    bool d_form = false;
    double tk_prev = 0.0;
    int done = 0;
    int64_t bytes_up = 0, bytes_down = 0, n_passes = 0;  // across PCIe (tvdn_run_stats)
    std::vector<void *> cdst, csrc;
    int down_blocks = 0;
    hipEvent_t pump_ready[4] = {nullptr, nullptr, nullptr, nullptr};  // "this chunk's rows are gathered", by chunk index mod 4 (two jobs are in flight at most)
    bool down_pump = false;  // downloads by the runtime's copies, one at a time from a helper thread (DownPump, tvdn_stream_parts.hpp)
    bool recon_direct = false, recon_direct_decided = false;  // the last pass sends the resident rows' results home itself
    bool lean_layout = false;  // every row kept, rings for the levels 1 .. K-1 only, no boxes (tvdn_stream.hip set_up)
    int inplace_kind = 0;  // kept rows swept in place (tvdn_stream_chain.hip): 0 no, 1 where their neighbours are kept too, 2 every one (all rows kept)
    std::vector<double> ratios;
    int ran = 0, ran_phase[2] = {0, 0};
    std::chrono::steady_clock::time_point t_passes, t_end_passes;
    double first_pass_s = 0.0;

    // ---- small helpers (the lambdas of round 4) -------------------------------------------------------------------------------------
    static double since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
    static size_t aligned(size_t b) { return (b + 255) / 256 * 256; }
    // chain(): recon crosses PCIe every pass (round 4's way) instead of being rebuilt from the state at level 0 (tvdn_rebuild.hip)
    bool ships_recon() const { return exact_wrap || a->use_stop || getenv("TVDN_STREAM_SHIP_RECON") != nullptr; }
    int depth_of_pass(int q) const { return a->use_stop ? 1 : n_total / n_pass_plan + (q < n_total % n_pass_plan ? 1 : 0); }
    bool resident(int64_t g) const { return RES > 0 && rm.resident(g); }
    char *host_row(const HostArr &h, int64_t g) const { return h.cube_rows ? h.p + (size_t)g * row_bytes : h.p + (size_t)rm.host_below(g) * row_bytes; }
    char *take(size_t b)
    {
        char *p = cursor;
        cursor += b;
        return p;
    }
    char *store_row(int i, int64_t g) const { return store[(size_t)i] + (size_t)rm.res_below(g) * row_bytes; }
    Ring &A(int64_t level, int q) { return Aw[(size_t)(level + 1) * nd + q]; }
    int wait_recon(int set) { return ((two_sets && set) ? recon2_ready : recon_ready).wait(); }
    int64_t local_slot(int64_t v) const { return (v - sh->local_v0) - rm.res_below(v - KX); }  // packed: resident rows have no slot
    // row of a host array by cube row g / virtual row v (a slab of its own process addresses its local arrays by v)
    char *hrow(const HostArr &h, int64_t g, int64_t v) const { return local_rows ? h.p + (size_t)local_slot(v) * row_bytes : host_row(h, g); }
    char *srow(int set, int arr, int64_t g, int64_t v) const
    {
        // (a slab of a device list: the shared arrays are indexed by cube row, kept rows or not)
        return local_rows ? sb[set].flat[(size_t)arr] + (size_t)local_slot(v) * row_bytes : sb[set].row(arr, sh ? g : rm.host_below(g));
    }
    int sse_row(const char *x, const char *y, int slot, int64_t g)
    {
        return tvdn_sum_square_error(ctx.c, a->dtype, nd, one_row, x, y, (double *)mse_d.p + (size_t)slot * (size_t)N0 + (size_t)g, st.main);
    }
    void pack_host_rows(char *packed, char *cube, bool to_packed);  // host rows of a cube-shaped user array <-> a packed buffer
    int wait_staged(int64_t upto);                                  // the resident rows below `upto` have their data term in the store
    int meet();                                                     // slabs of a device list: every pass ends at the barrier
    int stop_after(int slot, bool &stop);

    // ---- the steps -------------------------------------------------------------------------------------------------------------------
    int run();  // set_up, schedule, finish
    int set_up(bool &nothing_to_do);
    int pass(const double *ratios_of_pass /* kk entries, NAN = unaccelerated */, int kk);
    int describe(int it0, int kk, const double *rat, PassDesc &pd);
    int chain(std::vector<PassDesc> &ps);
    int schedule();
    int finish();
};

}  // namespace tvdn
