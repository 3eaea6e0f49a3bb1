// dsoft_engine.hpp -- host side of the device D-SOFT filter (gact_hip_dsoft_build / _query,
// gact_hip_candidates_download).  Not a header in its own right: included once by gact_engine.hip inside its
// extern "C" block, where the engine's internals (gact_hip_engine, Slot, SeqSet, DsoftState, HIP_TRY, fail) are
// in scope.  Kernels: dsoft_device.hpp.
// ---------------------------------------------------------------------------
// D-SOFT on the device

int gact_hip_dsoft_build(gact_hip_engine *e, const gact_dsoft_params *p, gact_dsoft_info *info)
{
    if (!e || !p) return fail(GACT_HIP_EINVAL, "dsoft_build: NULL argument");
    if (p->seed_size < 4 || p->seed_size > 15 || p->window_size < 1 || p->window_size >= p->seed_size)
        return fail(GACT_HIP_EINVAL, "dsoft_build: need 4 <= seed_size <= 15 and 1 <= window_size < seed_size "
                                     "(seed_pos_table.cpp:48-50)");
    if (p->bin_size < 1 || p->threshold < 1 || p->num_seeds < 0 || p->seed_occurence_multiple < 1)
        return fail(GACT_HIP_EINVAL, "dsoft_build: bin_size, threshold, seed_occurence_multiple must be >= 1");
    if (p->threshold + p->seed_size > 255)
        return fail(GACT_HIP_EINVAL, "dsoft_build: threshold + seed_size must stay under 256 (8-bit band counters)");
    std::lock_guard<std::mutex> lk(e->upload_mu);
    std::lock_guard<std::mutex> lk2(e->dsoft_mu);
    int rc = set_device(e);
    if (rc) return rc;
    const SeqSet &rs = e->sets[GACT_SET_REF];
    if (rs.n == 0 || !rs.d_raw) return fail(GACT_HIP_EINVAL, "dsoft_build: the reference read set has not been uploaded");
    Slot &sl = e->slots[0];
    DsoftState &d = e->dsoft;
    d.release_index();
    d.release_scratch();
    d.p = *p;

    // darwin.cpp:532-543: every sequence is padded with 'N' to a whole number of bins
    const uint32_t bin = (uint32_t)p->bin_size;
    std::vector<uint32_t> start_bin((size_t)rs.n);
    uint64_t cur = 0;
    for (int32_t i = 0; i < rs.n; i++) {
        start_bin[i] = (uint32_t)cur;
        const uint64_t len = (uint64_t)(rs.h_offsets[i + 1] - rs.h_offsets[i]);
        cur += (len + bin - 1) / bin;
        if (cur * bin > 0xfff00000ull) return fail(GACT_HIP_ERANGE, "dsoft_build: padded reference exceeds 32-bit positions");
    }
    d.n_bins_used = (uint32_t)cur;
    d.ref_len = (uint32_t)(cur * bin);
    if (d.ref_len < (uint32_t)(p->seed_size + p->window_size))
        return fail(GACT_HIP_EINVAL, "dsoft_build: reference shorter than one window");
    d.max_occ = (uint32_t)p->seed_occurence_multiple * (1u + (d.ref_len >> (2 * p->seed_size)));   // seed_pos_table.cpp:59
    if (p->max_candidates < 1) return fail(GACT_HIP_EINVAL, "dsoft_build: max_candidates must be >= 1");
    d.n_table = 1ull << (2 * p->seed_size);

    const uint32_t rlen_2bit = 1 + d.ref_len / 16;                                // :61
    const uint32_t end = 16 * rlen_2bit - (uint32_t)p->seed_size - (uint32_t)p->window_size;
    const uint32_t n_words = rlen_2bit + 3;
    const int n_mblocks = (int)((end + dsoft::kMinPerBlock - 1) / dsoft::kMinPerBlock);
    const int n_sblocks = (int)((d.n_table + dsoft::kScanPerBlock - 1) / dsoft::kScanPerBlock);

    int32_t *d_lastflag = nullptr, *d_carry = nullptr;
    uint32_t *d_sums = nullptr;
    unsigned long long *d_nemit = nullptr;
    auto cleanup = [&] {
        for (void *q : {(void *)d_lastflag, (void *)d_carry, (void *)d_sums, (void *)d_nemit}) if (q) (void)hipFree(q);
    };
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
#define DS_TRY(expr)                                                                       \
    do {                                                                                   \
        hipError_t err__ = (expr);                                                         \
        if (err__ != hipSuccess) {                                                         \
            cleanup(); d.release_index();                                                  \
            if (ev_a) (void)hipEventDestroy(ev_a);                                         \
            if (ev_b) (void)hipEventDestroy(ev_b);                                         \
            return fail(err__ == hipErrorOutOfMemory ? GACT_HIP_ENOMEM : GACT_HIP_EDEVICE, \
                        "%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, hipGetErrorString(err__)); \
        }                                                                                  \
    } while (0)
    DS_TRY(hipEventCreate(&ev_a));
    DS_TRY(hipEventCreate(&ev_b));
    DS_TRY(hipMalloc((void **)&d.d_start_bin, (size_t)rs.n * sizeof(uint32_t)));
    DS_TRY(hipMalloc((void **)&d.d_bin_chr, (size_t)d.n_bins_used * sizeof(int32_t)));
    DS_TRY(hipMalloc((void **)&d.d_ref2, (size_t)n_words * sizeof(uint32_t)));
    DS_TRY(hipMalloc((void **)&d.d_table, (size_t)d.n_table * sizeof(uint32_t)));
    DS_TRY(hipMalloc((void **)&d_lastflag, (size_t)n_mblocks * sizeof(int32_t)));
    DS_TRY(hipMalloc((void **)&d_carry, (size_t)n_mblocks * sizeof(int32_t)));
    DS_TRY(hipMalloc((void **)&d_sums, (size_t)n_sblocks * sizeof(uint32_t)));
    DS_TRY(hipMalloc((void **)&d_nemit, sizeof(unsigned long long)));
    DS_TRY(hipMemcpyAsync(d.d_start_bin, start_bin.data(), start_bin.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                          sl.stream));
    DS_TRY(hipEventRecord(ev_a, sl.stream));
    DS_TRY(hipMemsetAsync(d.d_ref2, 0, (size_t)n_words * sizeof(uint32_t), sl.stream));
    DS_TRY(hipMemsetAsync(d.d_table, 0, (size_t)d.n_table * sizeof(uint32_t), sl.stream));
    DS_TRY(hipMemsetAsync(d_nemit, 0, sizeof(unsigned long long), sl.stream));
    hipLaunchKernelGGL(dsoft::bin_chr_kernel, dim3((d.n_bins_used + 255) / 256), dim3(256), 0, sl.stream,
                       d.d_start_bin, rs.n, d.n_bins_used, d.d_bin_chr);
    const uint32_t data_words = (d.ref_len + 15) / 16;
    hipLaunchKernelGGL(dsoft::pack_ref_kernel, dim3((data_words + 255) / 256), dim3(256), 0, sl.stream,
                       rs.d_raw, rs.d_offsets, d.d_start_bin, d.d_bin_chr, bin, d.ref_len, d.d_ref2, data_words);
    hipLaunchKernelGGL(dsoft::ref_lastflag_kernel, dim3(n_mblocks), dim3(dsoft::kMinThreads), 0, sl.stream,
                       d.d_ref2, d.ref_len, end, p->seed_size, p->window_size, d_lastflag);
    hipLaunchKernelGGL(dsoft::carry_kernel, dim3(1), dim3(1024), 0, sl.stream, d_lastflag, n_mblocks, d_carry);
    hipLaunchKernelGGL(dsoft::ref_emit_kernel<0>, dim3(n_mblocks), dim3(dsoft::kMinThreads), 0, sl.stream,
                       d.d_ref2, end, p->seed_size, p->window_size, d_carry, d.d_table, (uint32_t *)nullptr, d_nemit);
    DS_TRY(hipGetLastError());
    unsigned long long n_emit = 0;
    DS_TRY(hipMemcpyAsync(&n_emit, d_nemit, sizeof n_emit, hipMemcpyDeviceToHost, sl.stream));
    DS_TRY(hipStreamSynchronize(sl.stream));
    d.n_min = (int64_t)n_emit;
    DS_TRY(hipMalloc((void **)&d.d_pos, (size_t)std::max<unsigned long long>(n_emit, 1) * sizeof(uint32_t)));
    hipLaunchKernelGGL(dsoft::scan_sums_kernel, dim3(n_sblocks), dim3(dsoft::kScanThreads), 0, sl.stream,
                       d.d_table, d.n_table, d_sums);
    hipLaunchKernelGGL(dsoft::scan_top_kernel, dim3(1), dim3(1024), 0, sl.stream, d_sums, n_sblocks);
    hipLaunchKernelGGL(dsoft::scan_apply_kernel, dim3(n_sblocks), dim3(dsoft::kScanThreads), 0, sl.stream,
                       d.d_table, d.n_table, d_sums);
    hipLaunchKernelGGL(dsoft::ref_emit_kernel<1>, dim3(n_mblocks), dim3(dsoft::kMinThreads), 0, sl.stream,
                       d.d_ref2, end, p->seed_size, p->window_size, d_carry, d.d_table, d.d_pos, d_nemit);
    hipLaunchKernelGGL(dsoft::sort_segments_kernel, dim3((unsigned)((d.n_table + 255) / 256)), dim3(256), 0, sl.stream,
                       d.d_table, d.n_table, d.max_occ, d.d_pos);
    DS_TRY(hipGetLastError());
    DS_TRY(hipEventRecord(ev_b, sl.stream));
    DS_TRY(hipStreamSynchronize(sl.stream));
    float ms = 0;
    DS_TRY(hipEventElapsedTime(&ms, ev_a, ev_b));
#undef DS_TRY
    cleanup();
    (void)hipEventDestroy(ev_a);
    (void)hipEventDestroy(ev_b);
    d.built = true;
    if (info) {
        memset(info, 0, sizeof *info);
        info->ref_length = d.ref_len; info->n_minimizers = d.n_min;
        info->table_bytes = (int64_t)d.n_table * 4; info->pos_bytes = d.n_min * 4;
        info->max_occurrence = (int32_t)d.max_occ; info->n_bins = (int32_t)d.n_bins_used;
        info->build_ms = ms;
    }
    return 0;
}

int gact_hip_dsoft_query(gact_hip_engine *e, int slot, int32_t first_query, int32_t n_queries, int32_t *n_forward,
                         int32_t *n_reverse, float *query_ms)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (!n_forward || !n_reverse) return fail(GACT_HIP_EINVAL, "dsoft_query: NULL argument");
    std::lock_guard<std::mutex> lk0(e->upload_mu);      // the sets must not change under the query (same order as dsoft_build)
    std::lock_guard<std::mutex> lk(e->dsoft_mu);
    DsoftState &d = e->dsoft;
    if (!d.built)
        return fail(GACT_HIP_EINVAL, "dsoft_query: no index for the resident GACT_SET_REF (gact_hip_dsoft_build has not "
                                     "been called since it was uploaded)");
    const SeqSet &rs = e->sets[GACT_SET_REF], &qf = e->sets[GACT_SET_QUERY], &qr = e->sets[GACT_SET_QUERY_RC];
    if (qf.n == 0 || qr.n != qf.n) return fail(GACT_HIP_EINVAL, "dsoft_query: both query sets must be uploaded");
    if (first_query < 0 || n_queries < 0 || (int64_t)first_query + n_queries > qf.n)
        return fail(GACT_HIP_ERANGE, "dsoft_query: queries [%d,%d) outside the set", first_query, first_query + n_queries);
    if (qf.max_len >= (1 << 24) || qr.max_len >= (1 << 24))
        return fail(GACT_HIP_ERANGE, "dsoft_query: reads of 16 Mb and more are not supported by the device filter");
    *n_forward = 0; *n_reverse = 0;
    if (query_ms) *query_ms = 0;
    if (n_queries == 0) return 0;
    if ((rc = set_device(e))) return rc;
    Slot &sl = e->slots[slot];
    const int n_tasks = 2 * n_queries;

    // scratch: one band-counter table per resident wave
    if (!d.d_tables) {
        const uint64_t worst = ((uint64_t)d.p.num_seeds + 1) * d.max_occ;           // hits of one query strand
        uint64_t size = 1024;
        while (size * 4 < worst * 5) size <<= 1;                                    // load factor <= 0.8 at the very worst
        if (size > (1ull << 26)) return fail(GACT_HIP_ENOMEM, "dsoft_query: band table of %llu slots per wave", (unsigned long long)size);
        const int blocks = e->prop.multiProcessorCount * 16;
        // a bin needs at least ceil(threshold / seed_size) seeds to cross the threshold
        const uint64_t per_cand = (uint64_t)((d.p.threshold + d.p.seed_size - 1) / d.p.seed_size);
        d.staged_cap = (uint32_t)(worst / std::max<uint64_t>(per_cand, 1) + 64);
        if (hipMalloc((void **)&d.d_tables, (size_t)blocks * size * sizeof(dsoft::BinSlot)) != hipSuccess ||
            hipMalloc((void **)&d.d_touched, (size_t)blocks * size * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void **)&d.d_staged, (size_t)blocks * d.staged_cap * sizeof(uint2)) != hipSuccess ||
            hipMalloc((void **)&d.d_next, 4 * sizeof(int)) != hipSuccess) {
            d.release_scratch();
            return fail(GACT_HIP_ENOMEM, "dsoft_query: scratch allocation failed");
        }
        HIP_TRY(hipMemsetAsync(d.d_tables, 0, (size_t)blocks * size * sizeof(dsoft::BinSlot), sl.stream));
        d.q_blocks = blocks;
        d.table_mask = (uint32_t)(size - 1);
    }
    // staging area of the candidates: a guess, corrected from the exact counts if it proves too small
    size_t temp_guess = (size_t)n_tasks * 32 + (1u << 20);
    if (const char *g = opt_env("dsoft_temp_cap")) temp_guess = (size_t)std::max(1L, atol(g));
    if (d.counts.reserve(n_tasks) || d.task_base.reserve(n_tasks) || d.out_base.reserve(n_tasks) ||
        d.temp.reserve(temp_guess))
        return fail(GACT_HIP_ENOMEM, "device allocation failed");

    dsoft::IndexDev ix;
    ix.ref2 = d.d_ref2; ix.table = d.d_table; ix.pos = d.d_pos; ix.bin_chr = d.d_bin_chr; ix.start_bin = d.d_start_bin;
    ix.ref_offsets = rs.d_offsets; ix.ref_len = d.ref_len; ix.n_bins_used = d.n_bins_used; ix.max_occ = d.max_occ;
    ix.k = d.p.seed_size; ix.w = d.p.window_size; ix.bin_size = (uint32_t)d.p.bin_size;
    ix.threshold = d.p.threshold; ix.num_seeds = d.p.num_seeds;
    const dsoft::QuerySetDev qfd{qf.d_packed, qf.d_offsets}, qrd{qr.d_packed, qr.d_offsets};
    const int blocks = std::min(n_tasks, d.q_blocks);
    dsoft::QueryScratch scr;
    scr.tables = d.d_tables; scr.table_mask = d.table_mask; scr.touched = d.d_touched; scr.staged = d.d_staged;
    scr.staged_cap = d.staged_cap;

    HIP_TRY(hipEventRecord(sl.ev0, sl.stream));
    std::vector<int32_t> counts((size_t)n_tasks);
    for (int attempt = 0;; attempt++) {
        dsoft::QueryOut qo;
        qo.counts = d.counts.p; qo.task_base = d.task_base.p;
        qo.temp_used = reinterpret_cast<unsigned long long *>(d.d_next + 2);
        qo.temp = d.temp.p; qo.temp_cap = (int64_t)d.temp.cap; qo.overflow = d.d_next + 1;
        HIP_TRY(hipMemsetAsync(d.d_next, 0, 4 * sizeof(int), sl.stream));
        hipLaunchKernelGGL(dsoft::query_kernel, dim3(blocks), dim3(64), 0, sl.stream, ix, qfd, qrd, first_query, n_queries,
                           d.d_next, scr, qo);
        HIP_TRY(hipGetLastError());
        int flags[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(counts.data(), d.counts.p, (size_t)n_tasks * sizeof(int32_t), hipMemcpyDeviceToHost, sl.stream));
        HIP_TRY(hipMemcpyAsync(flags, d.d_next, sizeof flags, hipMemcpyDeviceToHost, sl.stream));
        HIP_TRY(hipStreamSynchronize(sl.stream));
        if (!flags[1]) break;
        // the candidates did not fit the staging area: the counts are exact, size it from them and run again
        unsigned long long used = 0;
        memcpy(&used, &flags[2], sizeof used);
        if (attempt > 0 || d.temp.reserve((size_t)used + 1024))
            return fail(GACT_HIP_ENOMEM, "dsoft_query: %llu candidates do not fit the device", used);
    }
    // seed_pos_table.cpp:141-143: a query strand keeps its first max_candidates threshold crossings, in emission
    // order.  (What the reference's `break` does beyond that -- hits of the seed it skips, bins it never registers
    // for clearing and so carries into the NEXT query of the same host thread -- depends on which reads a thread
    // happens to get and is not reproduced: every query starts from clean bins here.)
    bool clamped = false;
    for (int t = 0; t < n_tasks; t++)
        if (counts[t] > d.p.max_candidates) { counts[t] = d.p.max_candidates; clamped = true; }
    if (clamped)
        HIP_TRY(hipMemcpyAsync(d.counts.p, counts.data(), (size_t)n_tasks * sizeof(int32_t), hipMemcpyHostToDevice, sl.stream));
    std::vector<int64_t> base((size_t)n_tasks);
    int64_t run = 0;
    for (int t = 0; t < n_tasks; t++) {
        if (t == n_queries) *n_forward = (int32_t)run;
        base[t] = run;
        run += counts[t];
    }
    if (run > 0x7fffffff) return fail(GACT_HIP_ERANGE, "dsoft_query: %lld candidates do not fit one launch", (long long)run);
    *n_reverse = (int32_t)run - *n_forward;
    const size_t n = (size_t)run;
    sl.n_cands = 0; sl.h_cands.clear(); sl.checked_key[0] = -1;
    if (reserve_candidates(sl, n))
        return fail(GACT_HIP_ENOMEM, "device allocation failed");
    HIP_TRY(hipMemcpyAsync(d.out_base.p, base.data(), (size_t)n_tasks * sizeof(int64_t), hipMemcpyHostToDevice, sl.stream));
    hipLaunchKernelGGL(dsoft::gather_kernel, dim3(std::min(n_tasks, 65536)), dim3(64), 0, sl.stream, d.temp.p,
                       d.task_base.p, d.counts.p, d.out_base.p, n_tasks, sl.cands.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.ev1, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    sl.n_cands = n;
    sl.cands_epoch = e->sets_epoch;          // the list is valid for these sets only (check_candidate_range)
    if (query_ms) HIP_TRY(hipEventElapsedTime(query_ms, sl.ev0, sl.ev1));
    return 0;
}

int gact_hip_candidates_download(gact_hip_engine *e, int slot, int32_t n, gact_candidate *out)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    Slot &sl = e->slots[slot];
    if (n < 0 || (size_t)n > sl.n_cands || (n > 0 && !out))
        return fail(GACT_HIP_EINVAL, "candidates_download: bad arguments (slot %d holds %zu candidates)", slot, sl.n_cands);
    if ((rc = set_device(e))) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(out, sl.cands.p, (size_t)n * sizeof(gact_candidate), hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    return 0;
}

