// StreamRun::pass -- one DRAINED pass of the streamed engine: kk iteration levels over the whole (virtual) cube, the pipeline filled
// and emptied inside the pass.  The form periodic cubes and slabs take (two sets of host state: a pass reads one and writes the
// other; artificial faces that give up a row per level; the exchange / all-reduce / row-0 hooks of a slab).  tvdn_stream.hip has
// the map of the engine.
#include "tvdn_stream_run.hpp"

namespace tvdn {

int StreamRun::pass(const double *ratios /* kk entries, NAN = unaccelerated */, int kk)
{
    std::vector<int> modes((size_t)kk);
    std::vector<double> tkp((size_t)kk);
    std::vector<char> forms((size_t)kk + 1);
    forms[0] = d_form;
    double prev = tk_prev;
    for (int j = 0; j < kk; ++j) {
        const bool acc = !std::isnan(ratios[j]);
        TVDN_REQUIRE(!acc || forms[j], "a FISTA iteration cannot follow an unaccelerated one");
        modes[j] = iter_mode(acc, forms[j] != 0);
        forms[j + 1] = acc;
        tkp[j] = prev;
        if (acc) prev = ratios[j];
    }
    const int n_in_state = forms[0] ? 2 : 1, n_out_state = forms[kk] ? 2 : 1;
    // The first pass of a run starts from recon = data term and all-zero accumulators (cyTVDN.py:131-145): neither is
    // uploaded -- the level-0 rows of recon are device copies of the data-term rows, those of the state copies of a plane
    // of zeros -- so the host arrays they will come down into need not exist yet.
    const bool first = n_passes == 0;
    if (sh && sh->before_pass && !first) {  // a slab of its own process: the neighbours' new rows into my halo rows
        const int rcb = sh->before_pass();
        if (rcb) return rcb;
    }
    // exact Jia-Zhao wrap across processes: the slab that owns row 0 sends row 0 of every level to the one that owns the
    // top face, once per pass (hooks of the caller; a slab in between has nothing to do with it)
    // Exact wrap over several slabs: the slab that owns row 0 hands row 0 of every level of the pass to EVERY other slab,
    // once per pass (a broadcast: every slab takes part).  Used by the slabs whose sweeps reach the cube's top face without
    // sweeping row 0 themselves -- the last slab, and any slab whose K-row halo reaches that far.
    const bool relay_send = exact_wrap && sh && sh->relay_row0 && sh->g0 == 0 && sh->g1 < N0;
    const bool relay_recv = exact_wrap && sh && sh->relay_row0 && sh->g0 > 0;
    int planes_ready = 0;
    bool relayed = false;
    // rows of the (virtual) cube this pass works on, and what each level can reach at an artificial face
    // (an artificial face -- the wrap of a periodic run, the face between two slabs -- gives up a row per level; a slab
    // whose halo would reach beyond a Jia-Zhao cube's own face stops at that face, which then is a real one)
    const int64_t E0 = art_lo ? (periodic ? own0 - kk : std::max(G0, own0 - kk)) : G0;
    const int64_t E1 = art_hi ? (periodic ? own1 + kk : std::min(G1, own1 + kk)) : G1;
    const bool shrink_lo = art_lo && (periodic || E0 > G0), shrink_hi = art_hi && (periodic || E1 < G1);
    auto lo_bound = [&](int64_t level) { return shrink_lo ? E0 + level : E0; };
    auto hi_bound = [&](int64_t level) { return shrink_hi ? E1 - level : E1; };
    const int64_t n_chunks = (E1 - E0 + kk + R - 1) / R;
    const int h_new = two_sets ? h_old ^ 1 : h_old;
    auto cube_row = [&](int64_t v) { return ((v - KX) % N0 + N0) % N0; };  // virtual row -> row of the cube

    // The host rows among the virtual rows [v0, v1) -> consecutive rows of a box, run by run: a run ends where the next
    // row is resident, where the cube wraps and where `contiguous(g, g + 1)` says the host memory is not in one piece.
    auto up_rows = [&](char *box, int64_t v0, int64_t v1, const std::function<char *(int64_t, int64_t)> &src_row,
                       const std::function<bool(int64_t)> &joins_next) -> int {
        int64_t slot = 0;
        for (int64_t v = v0; v < v1;) {
            const int64_t g = cube_row(v);
            if (resident(g)) {
                ++v;
                continue;
            }
            int64_t n = 1;
            while (v + n < v1 && g + n < N0 && !resident(g + n) && joins_next(g + n - 1)) ++n;
            TVDN_HIP(hipMemcpyAsync(box + (size_t)slot * row_bytes, src_row(g, v), (size_t)n * row_bytes, hipMemcpyHostToDevice, st.up));
            bytes_up += n * (int64_t)row_bytes;
            slot += n;
            v += n;
        }
        return TVDN_OK;
    };
    auto host_rows_in = [&](int64_t v0, int64_t v1) {
        int64_t n = 0;
        for (int64_t v = v0; v < v1; ++v) n += resident(cube_row(v)) ? 0 : 1;
        return n;
    };
    auto upload = [&](int64_t c) -> int {
        const int64_t u0 = E0 + c * R, u1 = std::min(E0 + (c + 1) * R, E1);
        if (u0 >= u1 || host_rows_in(u0, u1) == 0) return TVDN_OK;
        int rcu = orig_ready.wait();
        if (rcu) return rcu;
        const int h = (int)(c % 2);
        if (in_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.up, in_free[h], 0));
        auto joins = [&](const HostArr &ha) {  // in place: cube rows g and g+1 are adjacent; packed: host slots are
            return std::function<bool(int64_t)>([&ha](int64_t) { (void)ha; return true; });
        };
        int i = 0;
        if ((rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return hrow(orig_h, g, v); }, joins(orig_h)))) return rcu;
        if (!first) {
            if ((rcu = wait_recon(h_old))) return rcu;
            const HostArr &ro = (two_sets && h_old) ? recon2_h : recon_h;
            if ((rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return hrow(ro, g, v); }, joins(recon_h)))) return rcu;
            for (int q = 0; q < nd; ++q)
                for (int s = 0; s < n_in_state; ++s) {
                    const int arr = q * n_state + s;
                    if ((rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return srow(h_old, arr, g, v); },
                                       [&](int64_t g) { return sb[h_old].block_of(rm.host_below(g)) == sb[h_old].block_of(rm.host_below(g + 1)); })))
                        return rcu;
                }
        } else {
            i += 1 + nd * n_in_state;
        }
        if (want_mse && (rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return hrow(ref_h, g, v); }, joins(ref_h)))) return rcu;
        TVDN_HIP(hipEventRecord(in_ready[h], st.up));
        return TVDN_OK;
    };

    // downloads by the DMA engine, one copy at a time from a helper thread (DownPump, tvdn_stream_parts.hpp), as in chain()
    std::unique_ptr<DownPump> pump_holder;
    if (down_pump) {
        pump_holder.reset(new DownPump);
        pump_holder->start(device, st.down);
    }
    DownPump *pump = pump_holder.get();
    // consecutive launches of a level hand the axis-0 accumulator across their cut (as in chain(); every array of this pass is a
    // ring): ahead[j] = the row whose axis-0 output state level j has stored ahead, -1 = none.  TVDN_STREAM_HANDOVER=0: off.
    const char *e_chain = getenv("TVDN_STREAM_HANDOVER");
    const bool chainable = !(e_chain && atoi(e_chain) == 0);
    std::vector<int64_t> ahead((size_t)kk, -1);
    struct ChainOff {  // `it` outlives the pass: leave it as it was found
        tvdn_iter_args &it;
        ~ChainOff() { it.chain = 0; }
    } chain_off{it};
    int rc2 = upload(0);
    if (rc2) return rc2;
    for (int64_t c = 0; c < n_chunks; ++c) {
        if ((rc2 = upload(c + 1))) return rc2;  // the next chunk crosses PCIe while this one is swept
        const int h = (int)(c % 2);
        const int64_t u0 = E0 + c * R, u1 = std::min(E0 + (c + 1) * R, E1);
        if (u0 < u1) {
            const bool from_host = host_rows_in(u0, u1) > 0;
            if (from_host) TVDN_HIP(hipStreamWaitEvent(st.main, in_ready[h], 0));
            if (first && RES > 0 && (rc2 = wait_staged(std::min<int64_t>(N0, std::max<int64_t>(0, u1 - KX))))) return rc2;
            cdst.clear();
            csrc.clear();
            // row v of ring `rg` <- array `i_store` of the store (resident rows), box `i_box` (host rows, in their order),
            // or, in the first pass, the data-term row (recon) / the plane of zeros (state)
            auto scatter = [&](const Ring &rg, int i_store, int i_box, bool from_first, bool zeros) {
                int64_t slot = 0;
                for (int64_t v = u0; v < u1; ++v) {
                    const int64_t g = cube_row(v);
                    const bool res_row = resident(g);
                    const char *src;
                    if (from_first && zeros)
                        src = zero_plane;
                    else if (from_first)  // recon <- data term
                        src = res_row ? store_row(0, g) : inbox[h][0] + (size_t)slot * row_bytes;
                    else
                        src = res_row ? store_row(i_store, g) : inbox[h][i_box] + (size_t)slot * row_bytes;
                    if (!res_row) ++slot;
                    cdst.push_back(rg.row(v));
                    csrc.push_back((void *)src);
                }
            };
            int i = 0;
            scatter(Ow, 0, i++, false, false);
            scatter(Rw[0], 1, i++, first, false);
            for (int q = 0; q < nd; ++q) {
                scatter(A(0, q), 2 + q * n_state, i++, first, true);                            // level 0: d_k (or b)
                if (n_in_state == 2) scatter(A(-1, q), 2 + q * n_state + 1, i++, first, true);  // level -1: d_k-1
            }
            if (want_mse) scatter(Fw, -1, i++, false, false);
            rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
            if (rc2) return rc2;
            if (exact_wrap && u0 <= G0 && G0 < u1) {
                TVDN_HIP(hipMemcpyAsync(row0[0], Rw[0].row(G0), row_bytes, hipMemcpyDeviceToDevice, st.main));
                planes_ready = 1;
            }
            if (want_mse && first)  // MSE[0]: the input against the reference (cyTVDN.py:124-125), own rows
                for (int64_t g = std::max(u0, own0); g < std::min(u1, own1); ++g)
                    if ((rc2 = sse_row(Rw[0].row(g), Fw.row(g), 0, g - KX))) return rc2;
            if (from_host) {
                TVDN_HIP(hipEventRecord(in_free[h], st.main));
                in_free_set[h] = true;
            }
        }
        // the wavefront: level j+1 trails level j by one row
        for (int j = 0; j < kk; ++j) {
            const int64_t lo = std::max(lo_bound(j + 1), E0 + c * R - (j + 1)), hi = std::min(hi_bound(j + 1), E0 + (c + 1) * R - (j + 1));
            if (lo >= hi) continue;
            if (relay_recv && !relayed && hi == G1 && E0 > G0) {  // my first sweep at the cube's top face: row 0 of every level, from its owner
                int rcr = sh->relay_row0(0, row0_host.p, kk);
                if (rcr) {
                    set_error("the row-0 relay of a slab run failed (status %d)", rcr);
                    return TVDN_ERR_INVALID;
                }
                for (int q = 0; q < kk; ++q)
                    TVDN_HIP(hipMemcpyAsync(row0[(size_t)q], row0_host.p + (size_t)q * row_bytes, row_bytes, hipMemcpyHostToDevice, st.main));
                relayed = true;
            }
            it.mode = modes[j];
            it.tk = modes[j] == TVDN_ITER_FISTA_D ? ratios[j] : 0.0;
            it.tk_prev = tkp[j];
            it.recon_in = Rw[j].base;
            it.recon_out = Rw[j + 1].base;
            it.wrap_recon = exact_wrap ? row0[j] : nullptr;
            for (int q = 0; q < nd; ++q) {
                char *cur = A(j, q).base, *prv = A(j - 1, q).base, *nxt = A(j + 1, q).base;
                it.b_in[q] = it.d_in[q] = it.dprev_in[q] = nullptr;
                it.b_out[q] = it.d_out[q] = nullptr;
                if (modes[j] == TVDN_ITER_FISTA_D) {
                    it.d_in[q] = cur; it.dprev_in[q] = prv; it.d_out[q] = nxt;
                } else if (modes[j] == TVDN_ITER_FISTA_D_TO_PLAIN) {
                    it.d_in[q] = cur; it.dprev_in[q] = prv; it.b_out[q] = nxt;
                } else {
                    it.b_in[q] = cur; it.b_out[q] = nxt;
                }
            }
            // the sums count the cube's own rows once: wrapped rows (periodic) go to the discard slot
            const int64_t parts[3][2] = {{lo, std::min(hi, own0)}, {std::max(lo, own0), std::min(hi, own1)}, {std::max(lo, own1), hi}};
            for (int part = 0; part < 3; ++part) {
                const int64_t p0 = parts[part][0], p1 = parts[part][1];
                if (p0 >= p1) continue;
                it.sweep_lo = p0;
                it.sweep_hi = p1;
                it.chain = 0;
                if (chainable) {
                    if (ahead[(size_t)j] == p0 && p0 > it.row_lo) it.chain |= TVDN_SWEEP_CHAIN_LO;
                    // (the row stored ahead must not take the ring slot of the one row of this array that level j + 2 still reads as
                    // its d_k-1 in this chunk: the first row of its launch, when that launch is not chained -- see chain())
                    bool clobbers = false;
                    const int jj = j + 2;
                    if (jj < kk && (modes[(size_t)jj] == TVDN_ITER_FISTA_D || modes[(size_t)jj] == TVDN_ITER_FISTA_D_TO_PLAIN)) {
                        const int64_t s_lo = std::max(lo_bound(jj + 1), E0 + c * R - (jj + 1)), s_hi = std::min(hi_bound(jj + 1), E0 + (c + 1) * R - (jj + 1));
                        if (s_lo < s_hi && p1 - cap == s_lo) clobbers = !(ahead[(size_t)jj] == s_lo && s_lo > it.row_lo);
                    }
                    const bool store = p1 < it.row_hi && !clobbers;
                    if (store) it.chain |= TVDN_SWEEP_STORE_AHEAD;
                    ahead[(size_t)j] = store ? p1 : -1;
                }
                const int slot = part == 1 ? done + j : discard;
                rc2 = tvdn_iterate_fused(ctx.c, &it, (double *)sums_d.p + 3 * (size_t)slot, st.main);
                if (rc2) return rc2;
                if (want_mse && part == 1)
                    for (int64_t g = p0; g < p1; ++g)
                        if ((rc2 = sse_row(Fw.row(g), Rw[j + 1].row(g), done + j + 1, g - KX))) return rc2;
            }
            if (exact_wrap && lo == G0) {
                TVDN_HIP(hipMemcpyAsync(row0[j + 1], Rw[j + 1].row(G0), row_bytes, hipMemcpyDeviceToDevice, st.main));
                planes_ready = j + 2;
            }
        }
        if (relay_send && !relayed && planes_ready >= kk) {  // planes 0 .. kk-1 are what the top face's sweeps read
            for (int q = 0; q < kk; ++q)
                TVDN_HIP(hipMemcpyAsync(row0_host.p + (size_t)q * row_bytes, row0[(size_t)q], row_bytes, hipMemcpyDeviceToHost, st.main));
            TVDN_HIP(hipStreamSynchronize(st.main));
            const int rcr = sh->relay_row0(1, row0_host.p, kk);
            if (rcr) {
                set_error("the row-0 relay of a slab run failed (status %d)", rcr);
                return TVDN_ERR_INVALID;
            }
            relayed = true;
        }
        // rows that have reached the last level go home: resident rows into the store (device copies, in the same launch
        // as the gather of the others into the out box), the others across PCIe.  [lo, hi) are rows of the cube proper.
        const int64_t lo = std::max(own0, E0 + c * R - kk), hi = std::min(own1, E0 + (c + 1) * R - kk);
        if (lo < hi) {
            const bool to_host = host_rows_in(lo, hi) > 0;
            if (to_host && pump) {  // the box's last rows (two chunks ago) are home
                if ((rc2 = pump->wait(c - 2))) return rc2;
            } else if (to_host && out_free_set[h]) {
                TVDN_HIP(hipStreamWaitEvent(st.main, out_free[h], 0));
            }
            cdst.clear();
            csrc.clear();
            auto gather = [&](int i_store, int i_box, const Ring &rg) {
                int64_t slot = 0;
                for (int64_t v = lo; v < hi; ++v) {
                    const int64_t g = v - KX;
                    const bool res_row = resident(g);
                    cdst.push_back(res_row ? store_row(i_store, g) : outbox[h][i_box] + (size_t)slot * row_bytes);
                    csrc.push_back(rg.row(v));
                    if (!res_row) ++slot;
                }
            };
            int i = 0;
            gather(1, i++, Rw[kk]);
            for (int q = 0; q < nd; ++q) {
                gather(2 + q * n_state, i++, A(kk, q));
                if (n_out_state == 2) gather(2 + q * n_state + 1, i++, A(kk - 1, q));
            }
            rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
            if (rc2) return rc2;
            if (to_host) {
                DownPump::Job job;
                if (pump) {
                    job.id = c;
                    job.ready = pump_ready[c & 3];  // (the job of chunk c - 4 is long done: the box of chunk c - 2 has been waited for)
                    TVDN_HIP(hipEventRecord(job.ready, st.main));
                } else {
                    TVDN_HIP(hipEventRecord(out_ready[h], st.main));
                    TVDN_HIP(hipStreamWaitEvent(st.down, out_ready[h], 0));
                }
                auto down = [&](char *dst, const char *src, size_t len) -> int {
                    if (pump)
                        job.copies.push_back(DownPump::Copy{dst, src, len});
                    else
                        TVDN_HIP(hipMemcpyAsync(dst, src, len, hipMemcpyDeviceToHost, st.down));
                    return TVDN_OK;
                };
                if ((rc2 = wait_recon(h_new))) return rc2;  // the host arrays these rows land in exist (first pass: the helper may still be at it)
                const HostArr &rh = (two_sets && h_new) ? recon2_h : recon_h;
                // runs of host rows: consecutive cube rows (an array page-locked in place) or consecutive host slots inside
                // one block of host state -- a run must be one piece in every destination
                int64_t slot = 0;
                for (int64_t g = lo - KX; g < hi - KX;) {
                    if (resident(g)) {
                        ++g;
                        continue;
                    }
                    const int64_t hs = rm.host_below(g);
                    if ((rc2 = sb[h_new].wait_for(hs))) return rc2;
                    int64_t n = 1;
                    while (g + n < hi - KX && !resident(g + n) && sb[h_new].block_of(hs + n) == sb[h_new].block_of(hs)) ++n;
                    const size_t boff = (size_t)slot * row_bytes, len = (size_t)n * row_bytes;
                    i = 0;
                    if ((rc2 = down(hrow(rh, g, g + KX), outbox[h][i++] + boff, len))) return rc2;
                    for (int q = 0; q < nd; ++q)
                        for (int s = 0; s < n_out_state; ++s)
                            if ((rc2 = down(srow(h_new, q * n_state + s, g, g + KX), outbox[h][i++] + boff, len))) return rc2;
                    bytes_down += (int64_t)len * (1 + (int64_t)n_out_state * nd);
                    slot += n;
                    g += n;
                }
                if (pump) {
                    pump->push(std::move(job));
                } else {
                    TVDN_HIP(hipEventRecord(out_free[h], st.down));
                    out_free_set[h] = true;
                }
            }
        }
    }
    if (pump && (rc2 = pump->drain())) return rc2;
    if (relay_recv && !relayed) {  // a slab that had no use for the planes still takes part in the hand-over
        const int rcr = sh->relay_row0(0, row0_host.p, kk);
        if (rcr) {
            set_error("the row-0 relay of a slab run failed (status %d)", rcr);
            return TVDN_ERR_INVALID;
        }
        relayed = true;
    }
    TVDN_HIP(hipStreamSynchronize(st.down));
    TVDN_HIP(hipStreamSynchronize(st.main));
    TVDN_HIP(hipStreamSynchronize(st.up));
    d_form = forms[kk];
    tk_prev = prev;
    done += kk;
    h_old = h_new;
    ++n_passes;
    return TVDN_OK;
}

}  // namespace tvdn
