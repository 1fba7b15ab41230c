// StreamRun::chain -- the streamed engine's schedule for Jia-Zhao runs on one device: one pass, or several CHAINED into one
// running row index so that the link is busy both ways all the time.  tvdn_stream.hip has the map of the engine.
#include "tvdn_stream_run.hpp"

namespace tvdn {

int StreamRun::describe(int it0, int kk, const double *rat, PassDesc &pd)
{
    pd.it0 = it0;
    pd.kk = kk;
    pd.modes.resize((size_t)kk);
    pd.tk.resize((size_t)kk);
    pd.tkp.resize((size_t)kk);
    pd.first = n_passes == 0 && it0 == 0;
    pd.last = !a->use_stop && it0 + kk == n_total;
    bool form = d_form;
    pd.n_in_state = form ? 2 : 1;
    double prev = tk_prev;
    for (int j = 0; j < kk; ++j) {
        const bool acc = !std::isnan(rat[j]);
        TVDN_REQUIRE(!acc || form, "a FISTA iteration cannot follow an unaccelerated one");
        pd.modes[(size_t)j] = iter_mode(acc, form);
        pd.tk[(size_t)j] = acc ? rat[j] : 0.0;
        pd.tkp[(size_t)j] = prev;
        form = acc;
        if (acc) prev = rat[j];
    }
    pd.n_out_state = form ? 2 : 1;
    d_form = form;  // the trackers move on: the next description continues from here
    tk_prev = prev;
    return TVDN_OK;
}

// ---- chained passes (Jia-Zhao) ---------------------------------------------------------------------------------------------
// A pass of K levels over N0 rows fills and drains its pipeline: uploads happen in its first N0 / R chunks, downloads in
// its last N0 / R, and only the chunks in between use the link both ways -- at K = 38 on 64 rows that is 12 chunks of 52,
// although the link carries 56 GB/s up AND 49 GB/s down at once (tools/ubench/pcie_duplex.hip, profiles/r04_pcie_duplex.jsonl).
// Chained, pass p + 1 starts uploading in the chunk after pass p's last upload, while p's upper levels are still climbing:
// the passes are stacked into one running row index v = p N0 + g, level j + 1 trails level j by one row of THAT index, a
// ring slot is v mod ring, and a launch that straddles the seam between two passes is cut there (the last row of p ends
// at the cube's top face, row 0 of p + 1 starts at its bottom one; each piece with its own iteration numbers).  What pass
// p + 1 uploads must be home: row g comes down (p N0 + g + K) / R chunks in and goes up again ((p + 1) N0 + g) / R - 1
// chunks in, so K <= N0 - 3 R is asked for, and the upload stream waits for the download event of that chunk.
// One pass at a time (`chain` of one) is the drained schedule: runs with a stopping rule, and passes deeper than that.
// How rows come down.  The runtime's hipMemcpyAsync moves a download with a DMA engine when its stream is idle and with a
// blit kernel otherwise; two DMA transfers in opposite directions at once take 4 x as long (512 MiB down: 10.5 ms alone,
// 42 ms beside an upload; profiles/r04_chained_trace_summary.txt -- what made chained passes LOSE in round 3), while an
// upload by DMA beside a download by a small copy kernel runs at 54 + 42 GB/s and leaves the sweeps alone (8 workgroups;
// with 16 or more the sweeps lose a third: tools/ubench/pcie_duplex.hip, profiles/r04_pcie_duplex.jsonl).  Chained passes
// keep both directions busy all the time, so their downloads are a copy kernel of 8 workgroups writing the page-locked
// host arrays directly; drained passes keep the runtime's copies.  TVDN_STREAM_DOWN_BLOCKS=n overrides (0: runtime copies).
int StreamRun::chain(std::vector<PassDesc> &ps)
{
    const int P = (int)ps.size();
    const int64_t V1 = (int64_t)P * N0;  // running rows that are uploaded
    int64_t kmax = 0;
    for (const PassDesc &pd : ps) kmax = std::max<int64_t>(kmax, pd.kk);
    // Recon does not cross PCIe (tvdn_rebuild.hip): a pass that continues a run uploads the data term and the accumulator state,
    // and the level-0 recon of its host rows is REBUILT from them on the rings; only the run's last pass brings recon down.  The
    // last row of a chunk needs the axis-0 accumulator of the row after it, which is the first row of the NEXT chunk: the rebuild
    // reads those two planes straight from that chunk's in-box, i.e. chunk t is scattered once chunk t + 1 has arrived (its
    // upload started when chunk t - 1 left the box, a whole chunk's sweeps ago).  Shipped as before when the cube's first row is not finite (the exact
    // wrap keeps row 0's recon of every level, and its accumulator is not the constant zero the rebuild takes at the top face)
    // and with a stopping rule (any pass may be the last).
    const bool ship_recon = ships_recon();
    const int64_t n_chunks = (V1 + ps[(size_t)P - 1].kk + R - 1) / R;
    // downloads: the runtime's copies behind events (drained passes), a copy kernel of a few workgroups (down_blocks), or the
    // runtime's DMA engine one copy at a time from a helper thread (DownPump, tvdn_stream_parts.hpp)
    std::unique_ptr<DownPump> pump_holder;
    if (down_pump) {
        pump_holder.reset(new DownPump);
        pump_holder->start(device, st.down);
    }
    DownPump *pump = pump_holder.get();
    // Rows kept in HBM are swept IN PLACE where their neighbours are kept too (one pass at a time: P == 1, v == g).  The store of
    // the kept rows is a set of arrays; to a sweep an array is a ring longer than the cube (tvdn_iter_args.*_ring_rows), so level
    // 0 reads recon and the state straight from the store and the last level writes them there, instead of a copy of every
    // array into the level-0 rings and one out of the last level's.  In place are: level 0's inputs when every row it reads is
    // kept (not in a run's first pass, whose inputs are the data term and zeros); the last level's outputs and `orig` when every
    // row of the launch is kept; and when ALL rows are kept (`full`) also the state that two levels share -- level 1 reads d_k
    // from the store, level kk - 2 writes d_k+kk-1 there, level kk - 1 reads it there -- so that such a pass copies nothing.
    // Order makes it safe: the launches of a chunk run one after the other on one stream, level j trails level 0 by j rows, so
    // what level kk - 1 (kk - 2) overwrites at chunk t -- rows below (t + 1) R - kk (+ 1) -- level 0 (1) has read at chunks <= t
    // and never reads again: kk >= 2, and kk >= 3 where level kk - 2 writes the array level 0 reads (`full`).
    const PassDesc &pd0 = ps[0];
    const char *e_inplace = getenv("TVDN_STREAM_INPLACE");
    const bool inplace = RES > 0 && P == 1 && !want_mse && pd0.kk >= 2 && !(e_inplace && atoi(e_inplace) == 0);
    const bool full = inplace && RES == N0 && pd0.kk >= 3;
    // The lean layout (tvdn_stream.hip set_up) has no level-0 rings, no boxes and no data-term ring: also the run's FIRST pass is
    // in place -- level 0 reads recon from the data term in the store and its state from the plane of zeros (a ring of one row:
    // every row is that plane), level 1 its d_k-1 from the same plane.
    const bool lean = lean_layout;
    TVDN_REQUIRE(!lean || full, "the lean layout of a streamed run needs every pass swept in place (depth %d, %d pass(es) at once)", pd0.kk, P);
    inplace_kind = std::max(inplace_kind, lean ? 3 : (full ? 2 : (inplace ? 1 : 0)));
    const int64_t kAsArray = (1LL << 31) - 1;  // a ring size no row index reaches: slot = row
    auto all_kept = [&](int64_t g0, int64_t g1) {  // every cube row of [g0, g1), clipped to the cube, is kept (and there is one)
        g0 = std::max<int64_t>(g0, 0);
        g1 = std::min<int64_t>(g1, N0);
        return g0 < g1 && rm.host_below(g1) == rm.host_below(g0);
    };
    // array `i_store` of the store as an array indexed by cube row, valid over the run of kept rows around g (their slots are consecutive)
    auto kept_base = [&](int i_store, int64_t g) { return store_row(i_store, g) - (size_t)g * row_bytes; };
    auto is_d_mode = [](int m) { return m == TVDN_ITER_FISTA_D || m == TVDN_ITER_FISTA_D_TO_PLAIN; };
    // which kept rows still go through the level-0 rings: those a level-0 launch reads that also reads a streamed row
    std::vector<char> ring_in((size_t)(inplace ? N0 : 0), 0), ring_orig_row((size_t)(inplace ? N0 : 0), 0);
    if (inplace && !full)
        for (int64_t t = 0; t < n_chunks; ++t) {
            const int64_t v_lo = std::max<int64_t>(0, t * R - 1), v_hi = std::min<int64_t>(N0, (t + 1) * R - 1);
            if (v_lo < v_hi && !all_kept(v_lo - 1, v_hi + 1))
                for (int64_t g = std::max<int64_t>(v_lo - 1, 0); g < std::min<int64_t>(v_hi + 1, N0); ++g) ring_in[(size_t)g] = 1;
        }
    if (inplace && !full)  // (a launch of any level holds R consecutive rows: a kept row within R - 1 rows of a streamed one may share one with it)
        for (int64_t g = 0; g < N0; ++g) ring_orig_row[(size_t)g] = all_kept(g - (R - 1), g + R) ? 0 : 1;
    std::vector<hipEvent_t> down_done((size_t)n_chunks, nullptr);
    const int bx_recon = 1, bx_ref = 2 + nd * n_state;  // fixed box numbers: 0 data term, 1 recon, 2 + q n_state + s state
    auto bx_state = [&](int q, int s) { return 2 + q * n_state + s; };
    auto ox_state = [&](int q, int s) { return 1 + q * n_state + s; };  // out boxes: 0 recon, then the state
    it.shape[0] = V1;
    // the pieces of the running rows [v0, v1) by pass: fn(pass, v_lo, v_hi)
    auto pieces = [&](int64_t v0, int64_t v1, const std::function<int(int, int64_t, int64_t)> &fn) -> int {
        v0 = std::max<int64_t>(v0, 0);
        v1 = std::min<int64_t>(v1, V1);
        for (int64_t v = v0; v < v1;) {
            const int q = (int)(v / N0);
            const int64_t e = std::min<int64_t>(v1, (int64_t)(q + 1) * N0);
            const int rcp = fn(q, v, e);
            if (rcp) return rcp;
            v = e;
        }
        return TVDN_OK;
    };
    auto host_rows_in = [&](int64_t v0, int64_t v1) {
        int64_t n = 0;
        for (int64_t v = std::max<int64_t>(v0, 0); v < std::min(v1, V1); ++v) n += resident(v % N0) ? 0 : 1;
        return n;
    };
    // host rows among cube rows [g0, g1) -> box rows from `slot` on, run by run
    auto up_rows = [&](char *box, int64_t &slot, int64_t g0, int64_t g1, const std::function<char *(int64_t)> &src_row,
                       const std::function<bool(int64_t)> &joins_next) -> int {
        for (int64_t g = g0; g < g1;) {
            if (resident(g)) {
                ++g;
                continue;
            }
            int64_t n = 1;
            while (g + n < g1 && !resident(g + n) && joins_next(g + n - 1)) ++n;
            TVDN_HIP(hipMemcpyAsync(box + (size_t)slot * row_bytes, src_row(g), (size_t)n * row_bytes, hipMemcpyHostToDevice, st.up));
            bytes_up += n * (int64_t)row_bytes;
            slot += n;
            g += n;
        }
        return TVDN_OK;
    };
    const std::function<bool(int64_t)> always = [](int64_t) { return true; };
    auto upload = [&](int64_t t, int64_t t_now) -> int {
        const int64_t u0 = t * R, u1 = std::min((t + 1) * R, V1);
        if (u0 >= u1 || host_rows_in(u0, u1) == 0) return TVDN_OK;
        int rcu = orig_ready.wait();
        if (rcu) return rcu;
        const int h = (int)(t % 2);
        if (in_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.up, in_free[h], 0));
        int64_t s_orig = 0, s_recon = 0, s_ref = 0;
        std::vector<int64_t> s_state((size_t)nd * 2, 0);
        rcu = pieces(u0, u1, [&](int q, int64_t v_lo, int64_t v_hi) -> int {
            const PassDesc &pd = ps[(size_t)q];
            const int64_t g0 = v_lo - (int64_t)q * N0, g1 = v_hi - (int64_t)q * N0;
            const int64_t before = s_orig;
            int r3 = up_rows(inbox[h][0], s_orig, g0, g1, [&](int64_t g) { return host_row(orig_h, g); }, always);
            if (r3) return r3;
            const int64_t n_host = s_orig - before;
            if (want_mse && (r3 = up_rows(inbox[h][bx_ref], s_ref, g0, g1, [&](int64_t g) { return host_row(ref_h, g); }, always))) return r3;
            if (pd.first) {  // recon and state are formed on the device: their box rows stay unused
                s_recon += n_host;
                for (int64_t &x : s_state) x += n_host;
                return TVDN_OK;
            }
            if (q > 0) {
                // these rows came down at the end of pass q - 1: the upload stream waits for the chunk that sent the last of them
                const int64_t t_out = ((int64_t)(q - 1) * N0 + (g1 - 1) + ps[(size_t)q - 1].kk) / R;
                TVDN_REQUIRE(t_out < t_now && (pump || down_done[(size_t)t_out] != nullptr),
                             "chained passes: row %lld of pass %d is uploaded before pass %d has sent it home (k too deep to chain)",
                             (long long)(g1 - 1), q, q - 1);
                if (pump) {
                    if ((r3 = pump->wait(t_out))) return r3;
                } else {
                    TVDN_HIP(hipStreamWaitEvent(st.up, down_done[(size_t)t_out], 0));
                }
            }
            if (ship_recon) {
                if ((r3 = wait_recon(0))) return r3;
                if ((r3 = up_rows(inbox[h][bx_recon], s_recon, g0, g1, [&](int64_t g) { return host_row(recon_h, g); }, always))) return r3;
            } else {
                s_recon += n_host;  // rebuilt on the device: its box rows stay unused
            }
            for (int64_t g = g0; g < g1; ++g)
                if (!resident(g) && (r3 = sb[0].wait_for(rm.host_below(g)))) return r3;
            for (int qx = 0; qx < nd; ++qx)
                for (int s = 0; s < n_state; ++s) {
                    int64_t &sl = s_state[(size_t)qx * 2 + s];
                    if (s >= pd.n_in_state) {
                        sl += n_host;
                        continue;
                    }
                    const int arr = qx * n_state + s;
                    if ((r3 = up_rows(inbox[h][bx_state(qx, s)], sl, g0, g1, [&](int64_t g) { return sb[0].row(arr, rm.host_below(g)); },
                                      [&](int64_t g) { return sb[0].block_of(rm.host_below(g)) == sb[0].block_of(rm.host_below(g + 1)); })))
                        return r3;
                }
            return TVDN_OK;
        });
        if (rcu) return rcu;
        TVDN_HIP(hipEventRecord(in_ready[h], st.up));
        return TVDN_OK;
    };

    // The launches of a level follow each other row by row, chunk after chunk: each hands the axis-0 accumulator of the row
    // after its last to the next one (tvdn.h TVDN_SWEEP_STORE_AHEAD / _CHAIN_LO) instead of that one forming it again from
    // recon of the row before and the input state -- one plane less to read per launch, which is 19 -> 18 plane moves where a
    // launch is ONE row (the regime of a cube that streams every row: R 1, K 45-49).  Where every array of a level is a ring
    // for the whole pass, i.e. not where kept rows are swept in place (the store and the rings alternate there by launch).
    // ahead[j]: the running row whose axis-0 output state level j has stored ahead (-1: none).  TVDN_STREAM_HANDOVER=0: off.
    const char *e_chain = getenv("TVDN_STREAM_HANDOVER");
    const bool chainable = !inplace && !(e_chain && atoi(e_chain) == 0);
    std::vector<int64_t> ahead((size_t)kmax, -1);
    struct ChainOff {  // `it` outlives the call: leave it as it was found
        tvdn_iter_args &it;
        ~ChainOff() { it.chain = 0; }
    } chain_off{it};
    int rc2 = upload(0, 0);
    if (rc2) return rc2;
    for (int64_t t = 0; t < n_chunks; ++t) {
        if ((rc2 = upload(t + 1, t))) return rc2;  // the next chunk crosses PCIe while this one is swept
        const int h = (int)(t % 2);
        const int64_t u0 = t * R, u1 = std::min((t + 1) * R, V1);
        if (u0 < u1) {
            const bool from_host = host_rows_in(u0, u1) > 0;
            if (from_host) TVDN_HIP(hipStreamWaitEvent(st.main, in_ready[h], 0));
            cdst.clear();
            csrc.clear();
            int64_t slot = 0;
            for (int64_t v = u0; v < u1; ++v) {
                const int q = (int)(v / N0);
                const PassDesc &pd = ps[(size_t)q];
                const int64_t g = v - (int64_t)q * N0;
                const bool res_row = resident(g);
                if (pd.first && res_row && (rc2 = wait_staged(g + 1))) return rc2;
                auto put = [&](const Ring &rg, const char *src) {
                    cdst.push_back(rg.row(v));
                    csrc.push_back((void *)src);
                };
                auto boxed = [&](int bx) { return inbox[h][bx] + (size_t)slot * row_bytes; };
                const char *o_src = res_row ? store_row(0, g) : boxed(0);
                // a kept row goes through the rings only where a launch that reads it also reads a streamed row (above)
                const bool via_rings = !res_row || !inplace || (pd.first && !lean) || ring_in[(size_t)g];
                if (via_rings || ring_orig_row[(size_t)g]) put(Ow, o_src);
                if ((pd.first || res_row || ship_recon) && via_rings)  // (else: rebuilt from the state, below)
                    put(Rw[0], pd.first ? o_src : (res_row ? store_row(1, g) : boxed(bx_recon)));
                const bool level1_reads = !full && pd.kk >= 2 && is_d_mode(pd.modes[1]);  // d_k is level 1's d_k-1: read from its ring
                for (int qx = 0; qx < nd; ++qx) {
                    if (via_rings || level1_reads)
                        put(A(0, qx), pd.first ? zero_plane : (res_row ? store_row(2 + qx * n_state, g) : boxed(bx_state(qx, 0))));
                    if (pd.n_in_state == 2 && via_rings)
                        put(A(-1, qx), pd.first ? zero_plane : (res_row ? store_row(2 + qx * n_state + 1, g) : boxed(bx_state(qx, 1))));
                }
                if (want_mse) put(Fw, boxed(bx_ref));
                if (!res_row) ++slot;
            }
            rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
            if (rc2) return rc2;
            if (!ship_recon) {
                // level-0 recon of this chunk's host rows, pass by pass (each with its own form and momentum ratio), run of host
                // rows by run; the axis-0 accumulator of the row after a run: the next ring row, the store (a resident row), the
                // first row of the next chunk's box (its upload has been queued: a wait away), or nothing at a pass's top face
                bool next_box_waited = false;
                rc2 = pieces(u0, u1, [&](int q, int64_t v_lo, int64_t v_hi) -> int {
                    const PassDesc &pd = ps[(size_t)q];
                    if (pd.first) return TVDN_OK;  // recon = data term: copied above
                    RebuildArgs ra;
                    std::memset(&ra, 0, sizeof ra);
                    ra.dtype = a->dtype;
                    ra.ndim = nd;
                    for (int i = 1; i < nd; ++i) ra.plane_shape[i - 1] = a->shape[i];
                    ra.orig = Ow.base;
                    ra.recon = Rw[0].base;
                    ra.d_form = pd.n_in_state == 2;
                    for (int qx = 0; qx < nd; ++qx) {
                        ra.in1[qx] = ra.d_form ? A(-1, qx).base : A(0, qx).base;
                        ra.in2[qx] = ra.d_form ? A(0, qx).base : nullptr;
                        ra.lambda_mu[qx] = a->lambda_mu[qx];
                    }
                    ra.tk_prev = pd.tkp.empty() ? 0.0 : pd.tkp[0];
                    ra.top = (int64_t)(q + 1) * N0;
                    ra.ring_rows = cap;
                    ra.orig_ring_rows = ocap;
                    for (int64_t v = v_lo; v < v_hi;) {
                        if (resident(v - (int64_t)q * N0)) {  // its recon came from the store
                            ++v;
                            continue;
                        }
                        int64_t e = v + 1;
                        while (e < v_hi && !resident(e - (int64_t)q * N0)) ++e;
                        ra.row0 = v;
                        ra.row1 = e;
                        ra.next1 = ra.next2 = nullptr;
                        if (e == u1 && e < ra.top) {  // the row after this run has not been scattered: it belongs to the next chunk
                            const int64_t gn = e - (int64_t)q * N0;
                            const char *n1, *n2 = nullptr;  // axis 0: state 0 = d_k (or b), state 1 = d_k-1
                            if (resident(gn)) {
                                n1 = store_row(2, gn);
                                if (ra.d_form) n2 = store_row(3, gn);
                            } else {
                                if (!next_box_waited) TVDN_HIP(hipStreamWaitEvent(st.main, in_ready[(t + 1) % 2], 0));
                                next_box_waited = true;
                                n1 = inbox[(t + 1) % 2][bx_state(0, 0)];  // box row 0: the first host row of the next chunk is this very row
                                if (ra.d_form) n2 = inbox[(t + 1) % 2][bx_state(0, 1)];
                            }
                            ra.next1 = ra.d_form ? n2 : n1;
                            ra.next2 = ra.d_form ? n1 : nullptr;
                        }
                        const int r3 = recon_rebuild(ra, st.main);
                        if (r3) return r3;
                        v = e;
                    }
                    return TVDN_OK;
                });
                if (rc2) return rc2;
            }
            for (int64_t v = u0; v < u1; ++v) {
                const int q = (int)(v / N0);
                const int64_t g = v - (int64_t)q * N0;
                if (exact_wrap && g == 0) {
                    const bool in_ring = !resident(g) || !inplace || (ps[(size_t)q].first && !lean) || ring_in[(size_t)g];
                    const char *kept_recon = ps[(size_t)q].first ? store_row(0, g) : store_row(1, g);  // a run starts from recon = data term
                    TVDN_HIP(hipMemcpyAsync(((q & 1) ? row0b : row0)[0], in_ring ? Rw[0].row(v) : kept_recon, row_bytes, hipMemcpyDeviceToDevice, st.main));
                }
                if (want_mse && ps[(size_t)q].it0 == 0 && ps[(size_t)q].first)  // MSE[0]: the input against the reference (cyTVDN.py:124-125)
                    if ((rc2 = sse_row(Rw[0].row(v), Fw.row(v), 0, g))) return rc2;
            }
            if (from_host) {
                TVDN_HIP(hipEventRecord(in_free[h], st.main));
                in_free_set[h] = true;
            }
        }
        // the wavefront: level j+1 trails level j by one running row; a launch is cut at the seam between two passes
        for (int64_t j = 0; j < kmax; ++j) {
            rc2 = pieces(t * R - (j + 1), (t + 1) * R - (j + 1), [&](int q, int64_t v_lo, int64_t v_hi) -> int {
                const PassDesc &pd = ps[(size_t)q];
                if (j >= pd.kk) return TVDN_OK;
                const int mode = pd.modes[(size_t)j];
                it.row_lo = (int64_t)q * N0;
                it.row_hi = (int64_t)(q + 1) * N0;
                it.sweep_lo = v_lo;
                it.sweep_hi = v_hi;
                it.mode = mode;
                it.tk = pd.tk[(size_t)j];
                it.tk_prev = pd.tkp[(size_t)j];
                it.recon_in = Rw[(size_t)j].base;
                it.recon_out = Rw[(size_t)j + 1].base;
                it.wrap_recon = exact_wrap ? ((q & 1) ? row0b : row0)[(size_t)j] : nullptr;
                for (int qx = 0; qx < nd; ++qx) {
                    char *cur = A(j, qx).base, *prv = A(j - 1, qx).base, *nxt = A(j + 1, qx).base;
                    it.b_in[qx] = it.d_in[qx] = it.dprev_in[qx] = nullptr;
                    it.b_out[qx] = it.d_out[qx] = nullptr;
                    if (mode == TVDN_ITER_FISTA_D) {
                        it.d_in[qx] = cur; it.dprev_in[qx] = prv; it.d_out[qx] = nxt;
                    } else if (mode == TVDN_ITER_FISTA_D_TO_PLAIN) {
                        it.d_in[qx] = cur; it.dprev_in[qx] = prv; it.b_out[qx] = nxt;
                    } else {
                        it.b_in[qx] = cur; it.b_out[qx] = nxt;
                    }
                }
                it.orig = Ow.base;
                it.orig_ring_rows = ocap;
                it.recon_in_ring_rows = it.cur_ring_rows = it.prev_ring_rows = it.recon_out_ring_rows = it.out_ring_rows = 0;
                it.chain = 0;
                if (chainable) {
                    if (ahead[(size_t)j] == v_lo && v_lo > it.row_lo) it.chain |= TVDN_SWEEP_CHAIN_LO;
                    // The row stored ahead takes the ring slot of row v_hi - cap of level j + 1's axis-0 state.  A ring of R + 2 rows
                    // is sized for exactly what is read: level j + 2 reads that array as its d_k-1, and the one row of it that a
                    // full launch of level j would overwrite early is the first row of level j + 2's launch of THIS chunk -- read
                    // only by a prologue that is not chained (the first launch of a pass: rows from a face or a seam).  Then this
                    // launch keeps to its own rows, and the next one of the level forms its accumulator itself.
                    bool clobbers = false;
                    const int64_t w = v_hi - cap, jj = j + 2;
                    if (w >= 0 && w == t * R - (jj + 1)) {
                        const int qq = (int)(w / N0);
                        if (qq < P && jj < ps[(size_t)qq].kk && is_d_mode(ps[(size_t)qq].modes[(size_t)jj]))
                            clobbers = !(ahead[(size_t)jj] == w && w > (int64_t)qq * N0);
                    }
                    const bool store = v_hi < it.row_hi && !clobbers;
                    if (store) it.chain |= TVDN_SWEEP_STORE_AHEAD;
                    ahead[(size_t)j] = store ? v_hi : -1;
                }
                bool out_kept = false;
                if (inplace) {  // (P == 1: v == g)
                    const int64_t g_in = std::max<int64_t>(v_lo - 1, 0);
                    const bool in_kept = j == 0 && !pd.first && all_kept(v_lo - 1, v_hi + 1);
                    const bool from_zero = lean && pd.first;  // the first pass of a lean run: recon = data term, state = zeros
                    const bool rows_kept = all_kept(v_lo, v_hi);
                    out_kept = j == pd.kk - 1 && rows_kept;
                    const bool shared_kept = full && pd.n_out_state == 2;  // d_k+kk-1 goes to the store from level kk - 2, and level kk - 1 reads it there
                    if (in_kept) {
                        it.recon_in = kept_base(1, g_in);
                        it.recon_in_ring_rows = it.cur_ring_rows = it.prev_ring_rows = kAsArray;
                    } else if (from_zero && j == 0) {
                        it.recon_in = kept_base(0, g_in);
                        it.recon_in_ring_rows = kAsArray;
                        it.cur_ring_rows = it.prev_ring_rows = 1;
                    } else if (from_zero && j == 1 && is_d_mode(mode)) {
                        it.prev_ring_rows = 1;
                    }
                    if (out_kept) {
                        it.recon_out = kept_base(1, v_lo);
                        it.recon_out_ring_rows = it.out_ring_rows = kAsArray;
                    } else if (j == pd.kk - 2 && shared_kept) {
                        it.out_ring_rows = kAsArray;
                    }
                    if (j == 1 && full && !pd.first && is_d_mode(mode)) it.prev_ring_rows = kAsArray;
                    if (j == pd.kk - 1 && shared_kept) it.cur_ring_rows = kAsArray;
                    for (int qx = 0; qx < nd; ++qx) {
                        char *cur = nullptr, *prv = nullptr, *nxt = nullptr;
                        if (in_kept) {
                            cur = kept_base(2 + qx * n_state, g_in);
                            if (pd.n_in_state == 2) prv = kept_base(2 + qx * n_state + 1, g_in);
                        }
                        if (j == 1 && full && !pd.first && is_d_mode(mode)) prv = kept_base(2 + qx * n_state, g_in);
                        if (from_zero && j == 0) cur = prv = zero_plane;
                        if (from_zero && j == 1) prv = zero_plane;
                        if (j == pd.kk - 1 && shared_kept) cur = kept_base(2 + qx * n_state + 1, g_in);
                        if (out_kept)
                            nxt = kept_base(2 + qx * n_state, v_lo);
                        else if (j == pd.kk - 2 && shared_kept)
                            nxt = kept_base(2 + qx * n_state + 1, v_lo);
                        if (cur) (is_d_mode(mode) ? it.d_in[qx] : it.b_in[qx]) = cur;
                        if (prv && is_d_mode(mode)) it.dprev_in[qx] = prv;
                        if (nxt) (mode == TVDN_ITER_FISTA_D ? it.d_out[qx] : it.b_out[qx]) = nxt;
                    }
                    if (rows_kept && (!pd.first || lean)) {
                        it.orig = kept_base(0, v_lo);
                        it.orig_ring_rows = kAsArray;
                    }
                }
                int r3 = tvdn_iterate_fused(ctx.c, &it, (double *)sums_d.p + 3 * (size_t)(pd.it0 + (int)j), st.main);
                if (r3) return r3;
                if (want_mse)
                    for (int64_t v = v_lo; v < v_hi; ++v)
                        if ((r3 = sse_row(Fw.row(v), Rw[(size_t)j + 1].row(v), pd.it0 + (int)j + 1, v - (int64_t)q * N0))) return r3;
                if (exact_wrap && v_lo == (int64_t)q * N0)
                    TVDN_HIP(hipMemcpyAsync(((q & 1) ? row0b : row0)[(size_t)j + 1], out_kept ? store_row(1, 0) : Rw[(size_t)j + 1].row(v_lo), row_bytes, hipMemcpyDeviceToDevice,
                                            st.main));
                return TVDN_OK;
            });
            if (rc2) return rc2;
        }
        // rows that have reached their pass's last level go home: resident rows into the store, the others across PCIe
        cdst.clear();
        csrc.clear();
        struct Out {
            int q;
            int64_t g0, g1, slot0;
        };
        std::vector<Out> outs;
        int64_t oslot = 0;
        for (int q = 0; q < P; ++q) {
            const PassDesc &pd = ps[(size_t)q];
            const int64_t lo = std::max<int64_t>((int64_t)q * N0, t * R - pd.kk), hi = std::min<int64_t>((int64_t)(q + 1) * N0, (t + 1) * R - pd.kk);
            const bool recon_down = ship_recon || pd.last;  // a host row's recon goes home only when somebody will read it there
            if (lo >= hi) continue;
            // The run's last pass: the state of a resident row is not needed again, and its result can cross PCIe under
            // the pass (the link has room: a hybrid run uses half of it) instead of in one piece after it -- when the
            // caller's result array is page-locked in place, i.e. has a place for every row.
            if (pd.last && RES > 0 && !recon_direct_decided) {
                if ((rc2 = wait_recon(0))) return rc2;
                recon_direct = !lean && recon_h.cube_rows && getenv("TVDN_STREAM_HOME_AFTER") == nullptr;  // (the lean layout has no out box)
                recon_direct_decided = true;
            }
            const bool direct = pd.last && recon_direct;
            outs.push_back(Out{q, lo - (int64_t)q * N0, hi - (int64_t)q * N0, oslot});
            // (in place: these rows are the last level's launch of this chunk -- what it wrote into the store stays there)
            const bool out_kept = inplace && all_kept(lo, hi), shared_kept = full && pd.n_out_state == 2;
            for (int64_t v = lo; v < hi; ++v) {
                const int64_t g = v - (int64_t)q * N0;
                const bool res_row = resident(g);
                auto put = [&](int i_store, int ox, const Ring &rg) {
                    cdst.push_back(res_row ? store_row(i_store, g) : outbox[h][ox] + (size_t)oslot * row_bytes);
                    csrc.push_back(rg.row(v));
                };
                if (res_row && direct) {  // the result only, into the out box like a host row's
                    cdst.push_back(outbox[h][0] + (size_t)oslot * row_bytes);
                    csrc.push_back(out_kept ? (void *)store_row(1, g) : Rw[(size_t)pd.kk].row(v));
                    ++oslot;
                    continue;
                }
                if ((res_row || recon_down) && !out_kept) put(1, 0, Rw[(size_t)pd.kk]);
                for (int qx = 0; qx < nd; ++qx) {
                    if (!out_kept) put(2 + qx * n_state, ox_state(qx, 0), A(pd.kk, qx));
                    if (pd.n_out_state == 2 && !shared_kept) put(2 + qx * n_state + 1, ox_state(qx, 1), A(pd.kk - 1, qx));
                }
                if (!res_row) ++oslot;
            }
        }
        TVDN_REQUIRE(oslot <= R + 1, "chained passes: %lld rows come down in one chunk, the out boxes hold %lld (depths of consecutive passes differ by more than one)",
                     (long long)oslot, (long long)(R + 1));
        if (!cdst.empty()) {
            if (oslot > 0 && pump) {  // the box's last rows (two chunks ago) are home: the helper has waited for their copies
                if ((rc2 = pump->wait(t - 2))) return rc2;
            } else if (oslot > 0 && out_free_set[h]) {
                TVDN_HIP(hipStreamWaitEvent(st.main, out_free[h], 0));
            }
            rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
            if (rc2) return rc2;
        }
        if (oslot > 0) {
            DownPump::Job job;
            if (pump) {
                job.id = t;
                job.ready = pump_ready[t & 3];  // (the job of chunk t - 4 is long done: the box of chunk t - 2 has been waited for)
                TVDN_HIP(hipEventRecord(job.ready, st.main));
            } else {
                TVDN_HIP(hipEventRecord(out_ready[h], st.main));
                TVDN_HIP(hipStreamWaitEvent(st.down, out_ready[h], 0));
            }
            if ((rc2 = wait_recon(0))) return rc2;  // the host arrays these rows land in exist (first pass: the helper may still be at it)
            std::vector<void *> kd, ks;  // copies of whole rows for the copy kernel (down_blocks > 0)
            for (const Out &o : outs) {
                const PassDesc &pd = ps[(size_t)o.q];
                const bool direct = pd.last && recon_direct;
                int64_t slot = o.slot0;
                for (int64_t g = o.g0; g < o.g1;) {
                    if (resident(g)) {
                        if (direct) {  // its result went into the out box: one row, straight into the caller's array
                            char *dst = recon_h.p + (size_t)g * row_bytes;
                            const char *src = outbox[h][0] + (size_t)slot * row_bytes;
                            if (pump) {
                                job.copies.push_back(DownPump::Copy{dst, src, row_bytes});
                            } else if (down_blocks > 0 && row_bytes % 16 == 0 && ((uintptr_t)dst & 15) == 0) {
                                kd.push_back(dst);
                                ks.push_back((void *)src);
                            } else {
                                TVDN_HIP(hipMemcpyAsync(dst, src, row_bytes, hipMemcpyDeviceToHost, st.down));
                            }
                            bytes_down += (int64_t)row_bytes;
                            ++slot;
                        }
                        ++g;
                        continue;
                    }
                    const int64_t hs = rm.host_below(g);
                    if ((rc2 = sb[0].wait_for(hs))) return rc2;
                    int64_t n = 1;
                    while (g + n < o.g1 && !resident(g + n) && sb[0].block_of(hs + n) == sb[0].block_of(hs)) ++n;
                    const size_t boff = (size_t)slot * row_bytes, len = (size_t)n * row_bytes;
                    auto down = [&](char *dst, const char *src) -> int {
                        if (pump) {
                            job.copies.push_back(DownPump::Copy{dst, src, len});
                            return TVDN_OK;
                        }
                        if (down_blocks > 0 && row_bytes % 16 == 0 && ((uintptr_t)dst & 15) == 0) {
                            for (int64_t r = 0; r < n; ++r) {
                                kd.push_back(dst + (size_t)r * row_bytes);
                                ks.push_back((void *)(src + (size_t)r * row_bytes));
                            }
                        } else {
                            TVDN_HIP(hipMemcpyAsync(dst, src, len, hipMemcpyDeviceToHost, st.down));
                        }
                        return TVDN_OK;
                    };
                    const bool recon_down_q = ship_recon || pd.last;
                    if (recon_down_q && (rc2 = down(host_row(recon_h, g), outbox[h][0] + boff))) return rc2;
                    for (int qx = 0; qx < nd; ++qx)
                        for (int s = 0; s < pd.n_out_state; ++s)
                            if ((rc2 = down(sb[0].row(qx * n_state + s, hs), outbox[h][ox_state(qx, s)] + boff))) return rc2;
                    bytes_down += (int64_t)len * ((recon_down_q ? 1 : 0) + (int64_t)pd.n_out_state * nd);
                    slot += n;
                    g += n;
                }
            }
            // ONE launch of a few workgroups writes the chunk's rows into the page-locked host arrays (see down_blocks)
            if (!kd.empty() && (rc2 = tvdn_copy_many((int32_t)kd.size(), kd.data(), ks.data(), (int64_t)row_bytes, down_blocks, st.down))) return rc2;
            if (pump) {
                pump->push(std::move(job));
            } else {
                TVDN_HIP(hipEventRecord(out_free[h], st.down));
                out_free_set[h] = true;
            }
        }
        if (P > 1 && !pump) {  // what a later pass's uploads wait for (also for chunks that sent nothing: the stream is in order)
            if ((rc2 = evs.make(&down_done[(size_t)t]))) return rc2;
            TVDN_HIP(hipEventRecord(down_done[(size_t)t], st.down));
        }
    }
    if (pump && (rc2 = pump->drain())) return rc2;
    TVDN_HIP(hipStreamSynchronize(st.down));
    TVDN_HIP(hipStreamSynchronize(st.main));
    TVDN_HIP(hipStreamSynchronize(st.up));
    n_passes += P;
    return TVDN_OK;
}

}  // namespace tvdn
