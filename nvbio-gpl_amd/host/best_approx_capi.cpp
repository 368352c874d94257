// best_approx_capi.cpp -- the C++ host loop of host/nvbio_amd/best_approx.hpp behind one extern "C" entry point (libnvbio_amd_host.so), so that
// the parity tests and bench.py can drive it through ctypes.  Plain g++: no device code; everything on the GPU goes through libnvbio_amd.so.
#include <nvbio_amd/best_approx.hpp>
#include <cstring>

static thread_local char g_err[512] = "";

extern "C" {

struct nvbio_host_best_approx_params
{
    uint32_t seed_len, seed_freq, max_hits, rep_seeds, max_effort, max_effort_init, min_ext, max_ext, max_reseed, band, top_seed, batch_size, multi_hit;
};
struct nvbio_host_best_approx_stats { uint64_t n_extensions; uint32_t passes, multi_passes, seeding_passes, pad; };

const char* nvbio_host_last_error(void) { return g_err; }

// returns 0 on success; best_dev [4 n_reads] int32 (16-byte aligned), best_rc_dev [n_reads]
int nvbio_host_best_approx(int device, nvbio_fm_index_t fmi, const uint32_t* genome2_dev, uint32_t genome_len, const uint32_t* stored_reads4_dev,
                           const uint8_t* quals_dev, uint32_t n_reads, uint32_t read_len, int aln_type, const nvbio_gotoh_scheme* scheme, int32_t worst_score,
                           const nvbio_host_best_approx_params* p, int32_t* best_dev, uint8_t* best_rc_dev, void* stream, nvbio_host_best_approx_stats* stats)
{
    try
    {
        nvbio_amd_host::BestApproxParams q;
        q.seed_len = p->seed_len; q.seed_freq = p->seed_freq; q.max_hits = p->max_hits; q.rep_seeds = p->rep_seeds; q.max_effort = p->max_effort;
        q.max_effort_init = p->max_effort_init; q.min_ext = p->min_ext; q.max_ext = p->max_ext; q.max_reseed = p->max_reseed; q.band = p->band;
        q.top_seed = p->top_seed; q.batch_size = p->batch_size; q.multi_hit = p->multi_hit;
        const nvbio_amd_host::BestApproxStats s = nvbio_amd_host::best_approx( device, fmi, genome2_dev, genome_len, stored_reads4_dev, quals_dev, n_reads, read_len,
                                                                                (nvbio_alignment_type)aln_type, *scheme, worst_score, q, best_dev, best_rc_dev,
                                                                                (hipStream_t)stream );
        if (stats) { stats->n_extensions = s.n_extensions; stats->passes = s.passes; stats->multi_passes = s.multi_passes; stats->seeding_passes = s.seeding_passes; stats->pad = 0; }
        return 0;
    }
    catch (const std::exception& e)
    {
        strncpy( g_err, e.what(), sizeof(g_err) - 1 ); g_err[sizeof(g_err) - 1] = 0;
        return 1;
    }
}

struct nvbio_host_paired_params { uint32_t policy, min_frag_len, max_frag_len, overlap, unpaired; };
struct nvbio_host_paired_stats { uint64_t n_extensions, n_opposite; uint32_t passes, multi_passes; };

// the paired-end form: reads1 / reads2 stored reversed, uniform lengths read_len1 / read_len2; best_a_dev / best_o_dev [n_reads][2][4] int32
int nvbio_host_best_approx_paired(int device, nvbio_fm_index_t fmi, const uint32_t* genome2_dev, uint32_t genome_len, const uint32_t* stored_reads1_dev,
                                  const uint32_t* stored_reads2_dev, const uint8_t* quals1_dev, const uint8_t* quals2_dev, uint32_t n_reads, uint32_t read_len1,
                                  uint32_t read_len2, int aln_type, const nvbio_gotoh_scheme* scheme, int32_t worst_score1, int32_t worst_score2,
                                  const nvbio_host_best_approx_params* p, const nvbio_host_paired_params* pe, int32_t* best_a_dev, int32_t* best_o_dev, void* stream,
                                  nvbio_host_paired_stats* stats)
{
    try
    {
        nvbio_amd_host::BestApproxParams q;
        q.seed_len = p->seed_len; q.seed_freq = p->seed_freq; q.max_hits = p->max_hits; q.rep_seeds = p->rep_seeds; q.max_effort = p->max_effort;
        q.max_effort_init = p->max_effort_init; q.min_ext = p->min_ext; q.max_ext = p->max_ext; q.max_reseed = p->max_reseed; q.band = p->band;
        q.top_seed = p->top_seed; q.batch_size = p->batch_size; q.multi_hit = p->multi_hit;
        nvbio_amd_host::PairedParams pp; pp.policy = pe->policy; pp.min_frag_len = pe->min_frag_len; pp.max_frag_len = pe->max_frag_len; pp.overlap = pe->overlap; pp.unpaired = pe->unpaired;
        const uint32_t* reads[2] = { stored_reads1_dev, stored_reads2_dev }; const uint8_t* quals[2] = { quals1_dev, quals2_dev };
        const uint32_t lens[2] = { read_len1, read_len2 }; const int32_t worst[2] = { worst_score1, worst_score2 };
        const nvbio_amd_host::PairedStats s = nvbio_amd_host::best_approx_paired( device, fmi, genome2_dev, genome_len, reads, quals, n_reads, lens, (nvbio_alignment_type)aln_type,
                                                                                  *scheme, worst, q, pp, best_a_dev, best_o_dev, (hipStream_t)stream );
        if (stats) { stats->n_extensions = s.n_extensions; stats->n_opposite = s.n_opposite; stats->passes = s.passes; stats->multi_passes = s.multi_passes; }
        return 0;
    }
    catch (const std::exception& e)
    {
        strncpy( g_err, e.what(), sizeof(g_err) - 1 ); g_err[sizeof(g_err) - 1] = 0;
        return 1;
    }
}

} // extern "C"
