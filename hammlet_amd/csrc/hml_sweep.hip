// libhammlet_hip.so - the sweep for K states (one object per K: -DHML_TU_K=k, hammlet_amd/build.py): the kernels templated
// on the number of states and the host code that launches them, behind a table of function pointers (hml_ktab) that the
// core (hml_capi.hip) calls.  Reference: one Gibbs iteration, sampleHMM src/HMM.hpp:99-121 with
// StateSequence<ForwardBackward>::sample src/StateSequence/ForwardBackward.hpp:16-213 or <Mixture> Mixture.hpp:31-144.
#include "hml_capi_shared.hpp"

// Tile size and grid of the fused block kernel: the smallest number of 2^17-position batches per workgroup with which
// the whole grid is resident at once (the workgroups wait for lower-numbered ones inside the launch).  False when even
// the largest tile does not fit: those traces take the scan + scatter launches.
template <int KK>
static bool fused_geometry(hml_ctx* c, uint32_t* n_sub, uint32_t* n_wg) {
    if (c->fused_slots == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hml_k_blocks_fused<KK>, HML_FUSED_WAVES * 64, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || per_cu <= 0 || cus <= 0) {
            (void)hipGetLastError();
            c->fused_slots = -1;
        } else {
            c->fused_slots = per_cu * cus;
        }
        if (const char* e = getenv("HML_FUSED_SLOTS")) c->fused_slots = atoi(e);   // (tests: force larger tiles / an oversized grid)
    }
    if (c->fused_slots <= 0) return false;
    const uint64_t batches = (c->T + HML_FUSED_SUB_POSITIONS - 1) / HML_FUSED_SUB_POSITIONS;
    const uint64_t m = (batches + (uint64_t)c->fused_slots - 1) / (uint64_t)c->fused_slots;
    if (m > HML_FUSED_MAX_SUB) return false;
    *n_sub = (uint32_t)m;
    *n_wg = (uint32_t)((c->T + m * HML_FUSED_SUB_POSITIONS - 1) / (m * HML_FUSED_SUB_POSITIONS));
    return true;
}


template <int KK>
static int sweep_k(hml_ctx* c, char method, bool record) {
    hipStream_t s = c->stream;
    const bool mix = (method == HML_METHOD_MIXTURE);
    const uint32_t T = (uint32_t)c->T;
    bool emitted = false, fused = false;
    // The block count of the sweep sizes the grids and picks the forward geometry, and the host only knows the count of
    // an earlier sweep (it enqueues far ahead of the device).  Right after the parameters were replaced that count
    // means nothing - the first sweeps of a weakly compressed chain then ran in the geometry of a strongly compressed
    // one (177 ms instead of 19 ms each on C5, for as many sweeps as were enqueued at once) - so the block structure is
    // enumerated once ahead of the sweep and waited for.
    if (c->hint_stale && (c->dynamic || !c->blocks_valid)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            launch_compact_pair(c, 0, 0.0f);
            KLAUNCH_CHECK();
            HIPCHK(hipStreamSynchronize(s));
        }
    }
    c->hint_stale = false;
    // forward geometry of this sweep, fixed before its first launch
    refresh_hint(c);
    const bool dense_geo = c->B_hint >= c->dense_min_blocks;
    // weakly compressed univariate FB sweeps: emission terms, filter and candidate maps fused per tile (hml_k_trellis.h)
    const bool trellis = dense_geo && !mix && c->D == 1 && c->tre_fused;
    const bool mid_geo = !dense_geo && c->fwdL_mid > c->fwdL && c->B_hint >= c->mid_min_blocks;   // (hml_ctx.hpp: chunks of 8 from 2^18 blocks on)
    const int L = dense_geo ? c->fwdL_dense : mid_geo ? c->fwdL_mid : c->fwdL;
    const hml_layout lay = dense_geo ? c->lay_dense : mid_geo ? c->lay_mid : c->lay;
    // strongly compressed univariate sweeps keep no plane of rescale factors: the forward rows stay unscaled and the
    // backward maps apply the factor where they read a row (hml_bwd_row_load) - 3.5 of the block kernel's 11 MB of
    // stores at 10^8 positions, and what a dependent launch waits for is the write-back of its predecessor's stores
    float* const gsc_plane = (!dense_geo && c->late_rescale) ? nullptr : c->d_gsc;
    const uint32_t* const starts_for_maps = gsc_plane ? nullptr : c->d_starts;
    if (c->dynamic || !c->blocks_valid) {
        // the fused block kernel: univariate chains that have the GPU to themselves, unless compression is weak (the
        // float stream is the better access pattern then), the kernel reported a bounded wait that expired (someone
        // else is using the GPU: h_B[1]), or the trace is too long for a resident grid (hml_fused_geometry)
        uint32_t n_sub = 0u, n_wg = 0u;
        if (c->h_B[1] && !c->fused_keep) c->fused_blocks = false;
        if (c->D == 1 && c->use_keys && c->fused_blocks && !shares_device(c) &&
            !(!c->summary_always && c->B_hint && (uint64_t)c->B_hint * 24u > c->T) && fused_geometry<KK>(c, &n_sub, &n_wg)) {
            // K4 + K5 + K6a in one launch (hml_k_blocks_fused.h)
            ProfScope ps(c, "blocks_compact", 1);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_blocks_fused<KK>), dim3(n_wg), dim3(HML_FUSED_WAVES * 64), 0, s, c->d_summary, c->d_w, c->d_ia,
                               T, c->d_mdl, c->key_base, c->d_group_word, c->d_stage, c->d_starts, c->d_bstat, c->d_em,
                               gsc_plane, c->probes ? c->d_eprobe : nullptr, mix ? 1 : 0, lay, c->d_hB, n_sub, c->fused_spin_limit, c->d_dbg, c->d_mdl);
            fused = true;
        } else {
            launch_compact_pair(c, 0, 0.0f);
        }
        if (!fused && !trellis) {
            refresh_hint(c);
            const uint32_t h0 = c->B_hint ? c->B_hint : (uint32_t)std::min<uint64_t>(c->T, 1u << 20);
            ProfScope ps(c, "stats_emission");
            if (c->D > 1)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_emission_mv<KK, true>), dim3(grid_for(h0, 256, 64, 16384)), dim3(256), 0, s,
                                   c->d_ia, c->d_starts, c->d_mdl, c->d_bstat, c->d_em, gsc_plane, c->probes ? c->d_eprobe : nullptr,
                                   mix ? 1 : 0, lay);
            else if (dense_geo && L <= hml_emit_tile<KK>::MAXL)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_emission_tiled<KK, true>), dim3(grid_for(h0, hml_emit_tile<KK>::BLOCKS, 64, 65536)),
                                   dim3(256), 0, s, c->d_ia, c->d_starts, c->d_mdl, c->d_bstat, c->d_em, gsc_plane,
                                   c->probes ? c->d_eprobe : nullptr, mix ? 1 : 0, lay);
            else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_stats_emission<KK>), dim3(grid_for(h0, 256, 64, 16384)), dim3(256), 0, s,
                               c->d_ia, c->d_starts, c->d_mdl, c->d_bstat, c->d_em, gsc_plane, c->probes ? c->d_eprobe : nullptr,
                               mix ? 1 : 0, lay);
        }
        KLAUNCH_CHECK();
        emitted = true;
        if (!c->dynamic) c->blocks_valid = true;
    }
    refresh_hint(c);
    const uint32_t hint = c->B_hint ? c->B_hint : (uint32_t)std::min<uint64_t>(c->T, 1u << 20);
    const int gB = grid_for(hint, 256, 64, 16384);
    if (!emitted && !trellis) {
        ProfScope ps(c, "emission");
        if (c->D > 1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_emission_mv<KK, false>), dim3(gB), dim3(256), 0, s, c->d_ia, c->d_starts, c->d_mdl,
                               c->d_bstat, c->d_em, gsc_plane, c->probes ? c->d_eprobe : nullptr, mix ? 1 : 0, lay);
        else if (dense_geo && L <= hml_emit_tile<KK>::MAXL)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_emission_tiled<KK, false>), dim3(grid_for(hint, hml_emit_tile<KK>::BLOCKS, 64, 65536)),
                               dim3(256), 0, s, c->d_ia, c->d_starts, c->d_mdl, c->d_bstat, c->d_em, gsc_plane,
                               c->probes ? c->d_eprobe : nullptr, mix ? 1 : 0, lay);
        else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_emission<KK>), dim3(gB), dim3(256), 0, s, c->d_bstat, c->d_starts, c->d_mdl,
                           c->d_em, gsc_plane, c->probes ? c->d_eprobe : nullptr, mix ? 1 : 0, lay);
    }
    if (trellis) {
        // chunk length by the number of blocks: the warm-up (emission terms included) is paid once per chunk, and a
        // wavefront takes 64 chunks - long chunks where there are enough blocks to fill the machine with wavefronts anyway
        hipStreamCaptureStatus capst = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &capst);
        bool measure = false;
        if (c->tre_rows && c->tre_slots == 0) {   // wavefront slots of the first pass on this device (asked once)
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hml_k_trellis_rows<KK, false>, 64 * HML_TR2_WAVES, 0) == hipSuccess &&
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && per_cu > 0 && cus > 0)
                c->tre_slots = per_cu * cus * HML_TR2_WAVES;
            else { (void)hipGetLastError(); c->tre_slots = -1; }
            if (const char* e = getenv("HML_TRELLIS_SLOTS")) c->tre_slots = atoi(e);
        }
        const uint32_t TL = tre_pick_L(c, hint, capst != hipStreamCaptureStatusNone, &measure);
        hipEvent_t tev0 = nullptr, tev1 = nullptr;
        if (measure) { tev0 = ev_get(c); tev1 = ev_get(c); hipEventRecord(tev0, s); }
        if (c->graph_tre_L != TL && getenv("HML_TRELLIS_TUNE_DEBUG")) fprintf(stderr, "[trellis] chunk length %u for %u blocks (%d wavefront slots)\n", TL, hint, c->tre_slots);
        c->graph_tre_L = TL;
        const uint64_t tchunks = ((uint64_t)hint + TL - 1) / TL;
        const uint64_t tgroups = (tchunks + HML_TRE_NCH - 1) / HML_TRE_NCH;
        float* ep = c->probes ? c->d_eprobe : nullptr;
        float* ap = c->probes ? c->d_aprobe : nullptr;
        {
            ProfScope ps(c, "trellis", 1);
            // (nearly every block a single position: the filter step shares the candidate maps' sums, hml_k_trellis_rows.h)
            if (c->tre_rows && hint > 1024u && ((uint64_t)hint - 1024u) * 8u >= c->T * 9u)   // (last sweep's blocks >= 0.9 T; the hint carries 25 % headroom)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_rows<KK, true>), dim3(grid_for(tgroups, HML_TR2_WAVES, 4, 1 << 20)), dim3(64 * HML_TR2_WAVES), 0, s,
                                   c->d_ia, c->d_starts, c->d_mdl, c->d_mdl, c->d_bstat, c->d_smap, c->d_cmap, c->d_entry, c->d_exitA, c->d_fb, ep, ap, c->d_tre_ckpt, TL);
            else if (c->tre_rows)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_rows<KK, false>), dim3(grid_for(tgroups, HML_TR2_WAVES, 4, 1 << 20)), dim3(64 * HML_TR2_WAVES), 0, s,
                                   c->d_ia, c->d_starts, c->d_mdl, c->d_mdl, c->d_bstat, c->d_smap, c->d_cmap, c->d_entry, c->d_exitA, c->d_fb, ep, ap, c->d_tre_ckpt, TL);
            else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_tile<KK>), dim3(grid_for(tgroups, 1, 16, 1 << 20)), dim3(64), 0, s,
                               c->d_ia, c->d_starts, c->d_mdl, c->d_mdl, c->d_bstat, c->d_smap, c->d_cmap, c->d_entry, c->d_exitA, c->d_fb, ep, ap, TL);
        }
        {
            // verification, rounds of parallel refits from the predecessors' end vectors (two by default, hml_ctx.hpp; the lists of stale chunks
            // alternate between d_redo and d_redo2), then the sequential finisher: each exits at once when its list is empty
            ProfScope ps(c, "trellis_repair");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_verify<KK>), dim3(grid_for(tchunks, 256, 16, 1 << 14)), dim3(256), 0, s, c->d_mdl,
                               c->d_entry, c->d_exitA, c->d_redo, TL);
            int in_a = 1;
            for (uint32_t round = 0; round < c->tre_refit_rounds; ++round, in_a ^= 1) {
                uint32_t* lin = in_a ? c->d_redo : c->d_redo2;
                uint32_t* lout = in_a ? c->d_redo2 : c->d_redo;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_refit<KK>), dim3(4096), dim3(64), 0, s, c->d_ia, c->d_starts, c->d_mdl, c->d_smap,
                                   c->d_cmap, c->d_entry, c->d_exitA, c->d_fb, ep, ap, lin, in_a, (c->tre_rows && c->tre_ckpt) ? c->d_tre_ckpt : nullptr, TL);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_verify_list<KK>), dim3(64), dim3(256), 0, s, c->d_mdl, c->d_entry, c->d_exitA,
                                   lin, lout, c->d_touched, in_a, round, TL);
            }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_serial<KK>), dim3(1), dim3(256), 0, s, c->d_ia, c->d_starts, c->d_mdl, c->d_smap,
                               c->d_cmap, c->d_entry, c->d_exitA, c->d_fb, ep, ap, in_a ? c->d_redo : c->d_redo2, in_a, c->d_tre_bitmap, TL);
        }
        {
            ProfScope ps(c, "backward_chain");
            const uint64_t supers = (tchunks + 63) / 64;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_super<KK>), dim3(grid_for(supers * 64, 256, 16, 1 << 16)), dim3(256), 0, s, c->d_cmap,
                               c->d_mdl, c->d_scmap, c->d_super, TL);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_chain<KK>), dim3(1), dim3(1024), 0, s, c->d_super, c->d_mdl, c->d_bentry2, TL);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_trellis_states<KK>), dim3(grid_for(tgroups, 1, 16, 1 << 20)), dim3(64), 0, s, c->d_smap,
                               c->d_scmap, c->d_bentry2, c->d_mdl, c->d_q, TL);
        }
        if (measure) {
            // a measuring sweep (a few per chain): wait for the trellis kernels and note what this chunk length cost
            hipEventRecord(tev1, s);
            float ms = 0.0f;
            const bool ok = hipEventSynchronize(tev1) == hipSuccess && hipEventElapsedTime(&ms, tev0, tev1) == hipSuccess;
            c->ev_pool.push_back(tev0);
            c->ev_pool.push_back(tev1);
            if (ok) tre_tune_report(c, hint, ms); else { (void)hipGetLastError(); c->tre_autotune = false; }
        }
        c->tre_dense_sweeps++;
        {
            ProfScope ps(c, "counts");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts_dense<KK, false>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                               c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, (const unsigned long long*)nullptr, (const uint8_t*)nullptr);
        }
    } else if (!mix) {
        const uint64_t chunks = ((uint64_t)hint + L - 1) / L;
        const int gF = grid_for(chunks, 256, 16, 1 << 20);
        {
            ProfScope ps(c, "forward");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_forward<KK>), dim3(gF), dim3(256), 0, s, c->d_em, gsc_plane, c->d_mdl, c->d_rows,
                               c->probes ? c->d_aprobe : nullptr, c->d_entry, c->d_exitA, c->d_fb, L, lay);
        }
        {
            // backward maps (verifies the forward chunks on the way), then one workgroup: repair if a check failed,
            // and the chain over the chunk maps
            const uint64_t bch = ((uint64_t)hint + HML_BWD_CHUNK - 1) / HML_BWD_CHUNK;
            {
                ProfScope ps(c, "backward_maps");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_backward_maps<KK>), dim3(grid_for(bch * 64, 256, 16, 1 << 18)), dim3(256), 0,
                                   s, c->d_rows, c->d_mdl, c->d_smap, c->d_cmap, lay, c->d_entry, c->d_exitA, c->d_redo, L, starts_for_maps, c->d_mdl);
            }
            ProfScope ps(c, "backward_chain");
            if (!dense_geo) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_backward_chain<KK>), dim3(1), dim3(1024), 0, s, c->d_cmap, c->d_mdl,
                                   c->d_bentry, c->d_em, gsc_plane, c->d_rows, c->probes ? c->d_aprobe : nullptr, c->d_entry,
                                   c->d_exitA, c->d_fb, c->d_redo, c->d_touched, c->d_smap, L, lay, 3, 0, starts_for_maps);
            } else {
                // millions of backward chunks: repair step alone, then the two-level chain
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_backward_chain<KK>), dim3(1), dim3(1024), 0, s, c->d_cmap, c->d_mdl,
                                   c->d_bentry, c->d_em, gsc_plane, c->d_rows, c->probes ? c->d_aprobe : nullptr, c->d_entry,
                                   c->d_exitA, c->d_fb, c->d_redo, c->d_touched, c->d_smap, L, lay, 1, 0, starts_for_maps);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_backward_super<KK>), dim3(grid_for(bch, 256, 16, 1 << 16)), dim3(256), 0, s,
                                   c->d_cmap, c->d_mdl, c->d_scmap, c->d_super);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_backward_chain<KK>), dim3(1), dim3(1024), 0, s, c->d_super, c->d_mdl,
                                   c->d_bentry2, c->d_em, gsc_plane, c->d_rows, c->probes ? c->d_aprobe : nullptr, c->d_entry,
                                   c->d_exitA, c->d_fb, c->d_redo, c->d_touched, c->d_smap, L, lay, 2, 1, starts_for_maps);
                hipLaunchKernelGGL(hml_k_backward_entries, dim3(grid_for(bch, 256, 16, 1 << 16)), dim3(256), 0, s, c->d_scmap,
                                   c->d_bentry2, c->d_mdl, c->d_bentry);
            }
        }
        {
            ProfScope ps(c, "counts");
            if (c->D > 1)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts<KK, true, true>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                                   c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, c->d_smap, c->d_bentry);
            else if (dense_geo)   // hundreds of chunks per workgroup: one wavefront per chunk (same tree, bit for bit)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts_dense<KK, true>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                                   c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, c->d_smap, c->d_bentry);
            else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts<KK, true>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                               c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, c->d_smap, c->d_bentry);
        }
    } else {
        {
            ProfScope ps(c, "mixture");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_mixture<KK>), dim3(gB), dim3(256), 0, s, c->d_em, c->d_mdl, c->d_q, lay);
        }
        {
            ProfScope ps(c, "counts");
            if (c->D > 1)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts<KK, false, true>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                                   c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, (const unsigned long long*)nullptr,
                                   (const uint8_t*)nullptr);
            else if (dense_geo)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts_dense<KK, false>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                                   c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, (const unsigned long long*)nullptr,
                                   (const uint8_t*)nullptr);
            else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_counts<KK, false>), dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q,
                               c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, (const unsigned long long*)nullptr,
                               (const uint8_t*)nullptr);
        }
    }
    if (record && c->rec_marginals) {
        if (c->pooled) return set_err(HML_ERR_ARG, "the marginals of this context are pooled (common labels, several chains): further sweeps cannot be recorded into them");
        if (int r = ensure_marginal_buffers(c)) return r;
        ProfScope ps(c, "marginals");
        hipLaunchKernelGGL(hml_k_record, dim3(gB), dim3(256), 0, s, c->d_q, c->d_starts, c->d_mdl, c->d_diff, c->d_boundary);
    }
    {
        ProfScope ps(c, "params");
        if (c->params_spread) hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_params_spread<KK>), dim3(HML_PARAMS_TREE_WGS), dim3(1024), 0, s, c->d_mdl, c->d_partial);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_params<KK>), dim3(1), dim3(1024), 0, s, c->d_mdl, c->d_partial, 0);
    }
    KLAUNCH_CHECK();
    return 0;
}


template <int KK>
static int iterate_many_k(hml_ctx* const* cs, int n, uint64_t first, uint64_t iterations, uint64_t thinning, uint64_t* done) {
    hml_ctx* c0 = cs[0];
    hipStream_t s = c0->stream;
    const uint32_t T = (uint32_t)c0->T;
    const int L = c0->fwdL_many;
    const int with_gsc = c0->late_rescale ? 0 : 1;
    const uint32_t n_groups = (c0->n_spans + HML_GROUP_SPANS - 1) / HML_GROUP_SPANS;
    const bool records = thinning > 0 && thinning <= iterations;
    unsigned long long rec_mask = 0ull;
    for (int i = 0; i < n; ++i) {
        if (records && cs[i]->rec_marginals) {
            if (cs[i]->pooled) return set_err(HML_ERR_ARG, "the marginals of a context are pooled (common labels, several chains): further sweeps cannot be recorded into them");
            if (int r = ensure_marginal_buffers(cs[i])) return r;
            rec_mask |= 1ull << i;
        }
    }
    // the chains' pointers, in device memory of chain 0 (kept for the next call)
    std::vector<hml_chain_dev> h(n);
    for (int i = 0; i < n; ++i) {
        hml_ctx* c = cs[i];
        hml_chain_dev& d = h[i];
        d.summary = c->d_summary; d.w = c->d_w; d.ia = c->d_ia; d.key_base = c->key_base; d.n_spans = c->n_spans;
        d.stage = c->d_stage; d.span_count = c->d_span_count; d.coarse1 = c->d_coarse1; d.starts = c->d_starts; d.host_B = c->d_hB;
        d.bstat = c->d_bstat; d.mdl = c->d_mdl; d.em = c->d_em; d.gsc = c->d_gsc; d.rows = c->d_rows; d.entry = c->d_entry; d.exitv = c->d_exitA;
        d.fb = c->d_fb; d.redo = c->d_redo; d.touched = c->d_touched; d.smap = c->d_smap; d.cmap = c->d_cmap; d.bentry = c->d_bentry;
        d.q = c->d_q; d.partial = c->d_partial; d.diff = c->d_diff; d.boundary = c->d_boundary; d.lay = c->lay_many;
    }
    if (c0->many_cap < n) {
        if (c0->d_many) HIPCHK(hipFree(c0->d_many));
        c0->d_many = nullptr; c0->many_cap = 0;
        HIPCHK(hipMalloc(&c0->d_many, n * sizeof(hml_chain_dev)));
        c0->many_cap = n;
    }
    HIPCHK(hipMemcpyAsync(c0->d_many, h.data(), n * sizeof(hml_chain_dev), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // (the staging vector goes out of scope below)
    const hml_chain_dev* d_cs = (const hml_chain_dev*)c0->d_many;
    // The batch runs as G GROUPS of chains, each on a stream of its own (its first chain's), enqueued by this one host thread: a
    // group's latency-bound stretches - the one-workgroup kernels, the block kernel's hand-off, launch boundaries - are filled by
    // the other group's throughput-bound kernels (round 4, config 3, ms per round of sweeps: 2 chains 0.078 against 0.082, 4: 0.106 /
    // 0.126, 8: 0.158 / 0.174, 16: 0.259 / 0.307).  More than two groups lose again: the block kernel's pass over the summary and
    // the weights does not depend on the number of chains and is repeated per group, on tiles that grow with the number of grids
    // that have to be resident together (8 chains in 4 groups: 0.236).  HML_MANY_GROUPS overrides.
    int G = n >= 2 ? 2 : 1;
    if (const char* e = getenv("HML_MANY_GROUPS")) G = std::max(1, std::min(std::min(atoi(e), 4), n));
    struct Grp { int first, count; hipStream_t s; };
    std::vector<Grp> grp(G);
    for (int g = 0; g < G; ++g) { grp[g].first = (int)((int64_t)g * n / G); grp[g].count = (int)((int64_t)(g + 1) * n / G) - grp[g].first; grp[g].s = cs[grp[g].first]->stream; }
    // Chains attached to ONE trace (hml_attach_observations) take the many-chain block kernel: block starts, statistics and
    // emission terms of every chain from one pass over the shared summary / weights / integral array (hml_k_blocks_fused_many.h).
    // Its workgroups wait for lower-numbered ones inside the launch, so the whole grid must be resident - the tile grows
    // with the trace like the single-chain kernel's (fused_geometry).
    bool fm = c0->trace != nullptr && c0->fused_blocks;
    for (int k = 0; k < n; ++k) fm = fm && cs[k]->trace == c0->trace && cs[k]->fused_blocks && cs[k]->key_base == c0->key_base;
    uint32_t fm_sub = 0u, fm_wg = 0u;
    // HML_FM_SPLIT (default 1): the block structure of attached chains in two launches that no workgroup waits in
    // (hml_k_blocks_split_many.h) instead of the fused kernel; tiles of 2^17 positions (more per tile only beyond 4096 tiles)
    bool fm_split = true;
    if (const char* e = getenv("HML_FM_SPLIT")) fm_split = atoi(e) != 0;
    uint32_t fs_sub = 1u, fs_wg = 0u;
    const uint64_t fs_tiles1 = (c0->T + HML_FUSED_SUB_POSITIONS - 1) / HML_FUSED_SUB_POSITIONS;   // (tiles of one batch: what the per-tile arrays are sized for)
    {
        const uint64_t batches = (c0->T + HML_FUSED_SUB_POSITIONS - 1) / HML_FUSED_SUB_POSITIONS;
        fs_sub = (uint32_t)std::max<uint64_t>(1, (batches + 4095) / 4096);
        if (const char* e = getenv("HML_FM_SPLIT_SUB")) fs_sub = (uint32_t)std::max(1, atoi(e));   // (tests: tiles of several batches)
        if (fs_sub > HML_FUSED_MAX_SUB) fm_split = false;
        fs_wg = (uint32_t)((c0->T + (uint64_t)fs_sub * HML_FUSED_SUB_POSITIONS - 1) / ((uint64_t)fs_sub * HML_FUSED_SUB_POSITIONS));
    }
    if (fm && !fm_split) {
        if (c0->fm_slots == 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hml_m_blocks_fused<KK>, HML_FUSED_WAVES * 64, 0) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c0->device) != hipSuccess || per_cu <= 0 || cus <= 0) {
                (void)hipGetLastError();
                c0->fm_slots = -1;
            } else c0->fm_slots = per_cu * cus;
            if (const char* e = getenv("HML_FUSED_MANY_SLOTS")) c0->fm_slots = atoi(e);   // (tests: force larger tiles)
        }
        const uint64_t batches = (c0->T + HML_FUSED_SUB_POSITIONS - 1) / HML_FUSED_SUB_POSITIONS;
        // The block kernels of the G groups run side by side, and other contexts of the device that are not in this batch may be
        // running a batch of their own at the same time (another host thread): every launch takes its SHARE of the workgroup
        // slots, so that all these grids are resident together.
        const int live = c0->device < 64 ? g_live_ctx[c0->device].load() : n;
        const int sharers = G * std::max(1, (live + n - 1) / n);
        const int64_t slots = c0->fm_slots > 0 ? std::max<int64_t>(1, c0->fm_slots / sharers) : 0;
        const uint64_t m = slots > 0 ? (batches + (uint64_t)slots - 1) / (uint64_t)slots : HML_FUSED_MAX_SUB + 1;
        if (m > HML_FUSED_MAX_SUB) fm = false;
        else { fm_sub = (uint32_t)m; fm_wg = (uint32_t)((c0->T + m * HML_FUSED_SUB_POSITIONS - 1) / (m * HML_FUSED_SUB_POSITIONS)); }
    }
    // every chain's own stream goes on behind its group's (also on the early ways out)
    auto join = [&](bool wait) -> int {
        for (int g = 0; g < G; ++g) {
            if (wait) { HIPCHK(hipStreamSynchronize(grp[g].s)); continue; }
            hipEvent_t ev = ev_get(c0);
            HIPCHK(hipEventRecord(ev, grp[g].s));
            for (int k = grp[g].first; k < grp[g].first + grp[g].count; ++k) if (cs[k]->stream != grp[g].s) HIPCHK(hipStreamWaitEvent(cs[k]->stream, ev, 0));
            c0->ev_pool.push_back(ev);
        }
        return 0;
    };
    for (uint64_t i = first; i < iterations; ++i) {
        // a chain left the strongly compressed regime, or halted because its blocks outgrew its buffers (hml_state.h): back to the caller
        for (int k = 0; k < n; ++k) if (!many_sparse(cs[k]) || chain_halted(cs[k])) { *done = i; return join(true); }
        const bool record = thinning > 0 && ((i + 1) % thinning == 0);
        for (int k = 0; k < n; ++k) log_sweep(cs[k], HML_METHOD_FB, record);
        // (a bounded wait of the block kernel expired - somebody else is using the GPU: the scan + scatter launches from here on)
        if (fm) for (int k = 0; k < n; ++k) if (cs[k]->h_B[1] && !cs[k]->fused_keep) fm = false;
        for (int g = 0; g < G; ++g) {
            const int g0 = grp[g].first, gn = grp[g].count;
            hipStream_t s = grp[g].s;
            const hml_chain_dev* d_g = d_cs + g0;
            uint32_t hint = 0;
            for (int k = g0; k < g0 + gn; ++k) hint = std::max(hint, cs[k]->B_hint ? cs[k]->B_hint : (uint32_t)std::min<uint64_t>(T, 1u << 20));
            const unsigned ny = (unsigned)gn;
            const unsigned gB = (unsigned)grid_for(hint, 256, 64, 16384);
            if (fm && fm_split) {
                // two launches without a wait between workgroups (hml_k_blocks_split_many.h): the chains' block starts, then
                // statistics and emission terms - up to sixteen chains to a launch
                for (int k0 = g0; k0 < g0 + gn; k0 += HML_FS_MAX_CHAINS) {
                    const int nk = std::min(g0 + gn - k0, (int)HML_FS_MAX_CHAINS);
                    hml_fs_args fa;
                    memset(&fa, 0, sizeof fa);
                    for (int k = 0; k < nk; ++k) {
                        hml_ctx* c = cs[k0 + k];
                        hml_fs_chain& f = fa.c[k];
                        f.mdl = c->d_mdl; f.group_word = c->d_group_word; f.wave_total = c->d_wave_total; f.stage = c->d_stage; f.starts = c->d_starts;
                        f.tile_before = c->d_wave_total + (fs_tiles1 + 1) * HML_FUSED_WAVES; f.tile_prev = f.tile_before + (fs_tiles1 + 1);
                        f.bstat = c->d_bstat; f.em = c->d_em; f.gsc = with_gsc ? c->d_gsc : nullptr; f.host_words = c->d_hB; f.lay = c->lay_many;
                    }
                    hipLaunchKernelGGL(hml_m_blocks_list, dim3(fs_wg), dim3(HML_FUSED_WAVES * 64), 0, s, c0->d_summary, c0->d_w, T, c0->key_base, fa, nk, fs_sub);
                    hipLaunchKernelGGL(hml_m_blocks_offsets, dim3(1, (unsigned)nk), dim3(1024), 0, s, fa, fs_wg, fs_sub);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_blocks_emit<KK>), dim3(fs_wg), dim3(HML_FUSED_WAVES * 64), 0, s, c0->d_ia, T, fa, nk, fs_sub);
                }
            } else if (fm) {
                for (int k0 = g0; k0 < g0 + gn; k0 += HML_FM_MAX_CHAINS) {
                    const int nk = std::min(g0 + gn - k0, (int)HML_FM_MAX_CHAINS);
                    hml_fm_args fa;
                    memset(&fa, 0, sizeof fa);
                    for (int k = 0; k < nk; ++k) {
                        hml_ctx* c = cs[k0 + k];
                        hml_fm_chain& f = fa.c[k];
                        f.mdl = c->d_mdl; f.group_word = c->d_group_word; f.stage = c->d_stage; f.starts = c->d_starts; f.bstat = c->d_bstat;
                        f.em = c->d_em; f.gsc = with_gsc ? c->d_gsc : nullptr; f.host_words = c->d_hB; f.lay = c->lay_many;
                    }
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_blocks_fused<KK>), dim3(fm_wg), dim3(HML_FUSED_WAVES * 64), 0, s, c0->d_summary, c0->d_w, c0->d_ia,
                                       T, c0->key_base, fa, nk, fm_sub, c0->fused_spin_limit, g == 0 ? c0->d_dbg : nullptr);
                }
            } else {
                hipLaunchKernelGGL(hml_m_compact_scan_summary, dim3(n_groups, ny), dim3(256), 0, s, d_g, T);
                hipLaunchKernelGGL(hml_m_compact_scatter, dim3(n_groups, ny), dim3(256), 0, s, d_g, T);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_stats_emission<KK>), dim3(gB, ny), dim3(256), 0, s, d_g, with_gsc);
            }
            const uint64_t chunks = ((uint64_t)hint + L - 1) / L;
            const uint64_t bch = ((uint64_t)hint + HML_BWD_CHUNK - 1) / HML_BWD_CHUNK;
            // the chains' pointers travel as a kernel argument, up to sixteen chains to a launch (hml_many_args)
            for (int k0 = g0; k0 < g0 + gn; k0 += HML_MANY_ARG_CHAINS) {
                const int nk = std::min(g0 + gn - k0, (int)HML_MANY_ARG_CHAINS);
                hml_many_args ma;
                memset(&ma, 0, sizeof ma);
                for (int k = 0; k < nk; ++k) ma.c[k] = h[k0 + k];
                const unsigned nyk = (unsigned)nk;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_forward<KK>), dim3((unsigned)grid_for(chunks, 256, 16, 1 << 20), nyk), dim3(256), 0, s, ma, with_gsc, L);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_backward_maps<KK>), dim3((unsigned)grid_for((bch + 1) / 2 * 64, 256, 16, 1 << 18), nyk), dim3(256), 0, s, ma, with_gsc, L);   // (a wavefront per two chunks)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_backward_chain<KK>), dim3(1, nyk), dim3(1024), 0, s, ma, with_gsc, L);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_counts<KK>), dim3(HML_REDUCE_GROUPS, nyk), dim3(256), 0, s, ma);
                if (record && (rec_mask >> k0)) hipLaunchKernelGGL(hml_m_record, dim3(gB, nyk), dim3(256), 0, s, ma, rec_mask >> k0);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_m_params<KK>), dim3(HML_PARAMS_TREE_WGS, nyk), dim3(1024), 0, s, ma);
            }
        }
        KLAUNCH_CHECK();
        if (record) {
            bool any_cb = false;
            for (int k = 0; k < n; ++k) any_cb = any_cb || cs[k]->cb;
            if (any_cb) {
                if (int r = join(true)) return r;
                for (int k = 0; k < n; ++k) {
                    if (int r = check_device_error(cs[k])) return r;
                    if (cs[k]->cb && !chain_halted(cs[k])) cs[k]->cb(cs[k], i, cs[k]->cb_user);   // (a halted chain calls back when it catches up: hml_settle)
                }
            }
        }
    }
    if (int r = join(false)) return r;
    *done = iterations;
    return 0;
}



// ------------------------------------------------------------------------------------------------
// The K-dependent part behind its table (one per object: -DHML_TU_K=k; a development build: -DHML_ONLY_K=k; else all).
// (the tables are not `const`: the device pass of the compiler takes a constant with constant initialisers for a device
// constant as well and then looks for the host functions it points to; a plain host variable is only parsed there - which is
// what instantiates the kernels that sweep_k<K> launches)
#define HML_DEFINE_KTAB(KK)                                                                                                        \
    static void hml_kt_params_##KK(hml_ctx* c, int mode) {                                                                         \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_params<KK>), dim3(1), dim3(1024), 0, c->stream, c->d_mdl, c->d_partial, mode);    \
    }                                                                                                                              \
    static void hml_kt_derive_##KK(hml_ctx* c) {                                                                                   \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_derive<KK>), dim3(1), dim3(64), 0, c->stream, c->d_mdl);                          \
    }                                                                                                                              \
    extern hml_ktab hml_ktab_##KK;                                                                                                 \
    hml_ktab hml_ktab_##KK = {&sweep_k<KK>, &iterate_many_k<KK>, &hml_kt_params_##KK, &hml_kt_derive_##KK};
#if defined(HML_TU_K)
#define HML_DEFINE_KTAB_(K) HML_DEFINE_KTAB(K)
HML_DEFINE_KTAB_(HML_TU_K)
#elif defined(HML_ONLY_K)
#define HML_DEFINE_KTAB_(K) HML_DEFINE_KTAB(K)
HML_DEFINE_KTAB_(HML_ONLY_K)
#else
HML_DEFINE_KTAB(2) HML_DEFINE_KTAB(3) HML_DEFINE_KTAB(4) HML_DEFINE_KTAB(5) HML_DEFINE_KTAB(6) HML_DEFINE_KTAB(7) HML_DEFINE_KTAB(8) HML_DEFINE_KTAB(9)
HML_DEFINE_KTAB(10) HML_DEFINE_KTAB(11) HML_DEFINE_KTAB(12) HML_DEFINE_KTAB(13) HML_DEFINE_KTAB(14) HML_DEFINE_KTAB(15) HML_DEFINE_KTAB(16)
#endif
