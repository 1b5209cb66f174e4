// tests/emu/emu.cpp -- TEST-ONLY host stepping harness for the device code.  NOT part of the product and never
// loaded by the arriba_amd package.
//
// The build container has no GPU, so the per-fragment device functions (arriba_amd/csrc/device/*_core.hpp, all
// __host__ __device__) are also compiled here for the host and driven by plain loops, with the same call
// sequence as the HIP orchestration in agpu_api.hip.  This lets the CPU-only test tier check the kernel logic
// against the reference's golden dumps before any GPU minute is spent.  It exports the same entry points as
// include/arriba_gpu.h with the prefix emu_ instead of agpu_.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/arriba_gpu.h"
#include "../../arriba_amd/csrc/device/filter_core.hpp"
#include "../../arriba_amd/csrc/device/fusion_core.hpp"
#include "../../arriba_amd/csrc/device/evalue_host.hpp"
#include "../../arriba_amd/csrc/device/order_host.hpp"
#include "../../arriba_amd/csrc/device/mismapper_core.hpp"
#include "../../arriba_amd/csrc/device/merge_core.hpp"
#include "../../arriba_amd/csrc/device/event_core.hpp"
#include "../../arriba_amd/csrc/device/in_vitro_host.hpp"
#include "../../arriba_amd/csrc/device/homolog_host.hpp"
#include "../../arriba_amd/csrc/device/range_rule_host.hpp"
#include "../../arriba_amd/csrc/device/genomic_support_core.hpp"
#include "../../arriba_amd/csrc/device/index_bins.hpp"
#include "../../arriba_amd/csrc/device/multimapper_core.hpp"
#include "../../arriba_amd/csrc/device/ingest_core.hpp"
#include "../../arriba_amd/csrc/device/shard_host.hpp"
#include "../../arriba_amd/csrc/device/crc32_core.hpp"
#include "../../arriba_amd/csrc/device/inflate_fast_core.hpp"
#include <map>
#include <mutex>
#include <set>
#include <tuple>

using namespace agpu;

namespace {
std::string g_error;
const uint32_t MAX_SAMPLES = 100001;

double binomial_coefficient(const unsigned int k, const unsigned int n) {
	double result = 1;
	for (unsigned int i = n - k + 1; i <= n; ++i) result *= i;
	for (unsigned int i = 1; i <= k; ++i) result /= i;
	return result;
}
float binomial_distribution(const unsigned int k, const unsigned int n, const float p) { return binomial_coefficient(k, n) * pow(p, k) * pow(1 - p, n - k); }
bool mismatch_verdict(unsigned int mismatches, unsigned int alignment_length, const float mismatch_probability, unsigned long int genome_size, const float pvalue_cutoff) {
	if (binomial_distribution(mismatches, alignment_length, mismatch_probability) < pvalue_cutoff) return true;
	else if (mismatches > 0) {
		long double number_of_permutations_of_bases = pow(4, alignment_length - mismatches);
		if (genome_size >= number_of_permutations_of_bases) return true;
		return (1 - pow(1 - genome_size / number_of_permutations_of_bases, binomial_coefficient(mismatches, alignment_length))) > 0.01;
	}
	return false;
}
}

struct emu_ctx {
	agpu_params params;
	// annotation (host copies; gene table grows by the dummy genes)
	std::vector<uint16_t> gene_contig; std::vector<int32_t> gene_start, gene_end, gene_exonic_length; std::vector<uint8_t> gene_bits;
	std::vector<uint64_t> dummy_start_key, dummy_end_key;
	agpu_annotation_view in_annotation;
	AnnotationView annotation;
	agpu_genome_view in_genome;
	GenomeView genome;
	uint64_t genome_size = 0;
	// batch
	uint64_t n = 0;
	std::vector<uint8_t> fbits, filter, abits[3], gene_count[3];
	std::vector<uint32_t> genes[3], gene_pool;
	uint32_t gene_pool_used = 0;
	BatchView batch;
	uint32_t max_read_length = 0;
	std::vector<uint32_t> viral_pairs;
	std::vector<uint32_t> mismatch_table, kmer_table;
	std::vector<uint8_t> verdict_top, verdict_low;
	FilterTables tables;
	uint64_t stage_counts[16];
	std::vector<FusionEmission> emissions;
	uint32_t n_real_genes = 0;
	std::vector<uint64_t> unmapped;              // positions that need a dummy gene (between annotate_begin and annotate_finish)
	std::vector<uint8_t> duplicate_entries;      // wire format of the sharded duplicate exchange: 12-byte key + 4-byte global name rank
	uint64_t global_n = 0;
	// the reads of the sample sharded over the ranks (emu_shard_keep, emu_sharded.inc): what the walks over read lists ask of every fragment of the SAMPLE
	bool read_sharded = false, state_imported = false;
	std::vector<uint8_t> global_filter, global_bits; // [global_n]; bits: WALK_MULTIMAPPER | WALK_EXONIC
	std::vector<uint32_t> sample_gene_read_counts; bool have_sample_gene_read_counts = false; // emu_set_gene_read_counts
	std::vector<uint32_t> clipped_entries; // emu_in_vitro_clipped_mates: triples (candidate, count1, count2)
	std::vector<uint32_t> exon_bins, gene_bins;
	CoverageView coverage = { 0, nullptr, nullptr, nullptr, nullptr };
};

static void refresh_annotation(emu_ctx* ctx) {
	AnnotationView& a = ctx->annotation;
	const agpu_annotation_view& in = ctx->in_annotation;
	a.n_genes = ctx->n_real_genes; a.n_dummy = ctx->dummy_start_key.size();
	a.gene_contig = ctx->gene_contig.data(); a.gene_start = ctx->gene_start.data(); a.gene_end = ctx->gene_end.data(); a.gene_bits = ctx->gene_bits.data(); a.gene_exonic_length = ctx->gene_exonic_length.data();
	a.n_exons = in.n_exons; a.exon_start = in.exon_start; a.exon_end = in.exon_end; a.exon_gene = in.exon_gene; a.exon_previous = in.exon_previous; a.exon_next = in.exon_next;
	a.exon_cds_start = in.exon_cds_start; a.exon_cds_end = in.exon_cds_end;
	a.exon_index.n_contigs = in.exon_index.n_contigs; a.exon_index.contig_offset = in.exon_index.contig_offset; a.exon_index.keys = in.exon_index.keys; a.exon_index.member_offset = in.exon_index.member_offset; a.exon_index.members = in.exon_index.members;
	// the same directory over the keys as on the device (index_bins.hpp), so that the host stepping covers the narrowed binary search
	build_index_bins(in.exon_index.n_contigs, in.exon_index.contig_offset, in.exon_index.keys, ctx->exon_bins);
	a.exon_index.bins = ctx->exon_bins.data();
	build_index_bins(in.gene_index.n_contigs, in.gene_index.contig_offset, in.gene_index.keys, ctx->gene_bins);
	a.gene_index.bins = ctx->gene_bins.data();
	a.gene_index.n_contigs = in.gene_index.n_contigs; a.gene_index.contig_offset = in.gene_index.contig_offset; a.gene_index.keys = in.gene_index.keys; a.gene_index.member_offset = in.gene_index.member_offset; a.gene_index.members = in.gene_index.members;
	a.dummy_start_key = ctx->dummy_start_key.data(); a.dummy_end_key = ctx->dummy_end_key.data();
}

static void build_tables(emu_ctx* ctx) {
	const uint32_t max_length = std::max<uint32_t>(ctx->max_read_length, 1) + 8;
	ctx->mismatch_table.assign(((size_t) (max_length + 1) * (max_length + 2) / 2 + 31) / 32, 0);
	for (uint32_t n = 0; n <= max_length; ++n)
		for (uint32_t k = 0; k <= n; ++k)
			if (mismatch_verdict(k, n, (float) 0.01, ctx->genome_size, ctx->params.mismatch_pvalue_cutoff)) {
				uint32_t bit = n * (n + 1) / 2 + k;
				ctx->mismatch_table[bit >> 5] |= 1u << (bit & 31);
			}
	ctx->kmer_table.resize(max_length + 1);
	const unsigned int kmer_length = 3;
	const float kmer_content = ctx->params.max_kmer_content;
	for (uint32_t length = 0; length <= max_length; ++length) { unsigned int value = length * kmer_content / kmer_length + 0.5; ctx->kmer_table[length] = value; }
	FilterTables& t = ctx->tables;
	t.mismatch_verdict = ctx->mismatch_table.data(); t.mismatch_max_length = max_length;
	t.kmer_threshold = ctx->kmer_table.data(); t.kmer_threshold_size = max_length + 1; t.max_kmer_content = kmer_content;
	t.homopolymer_length = ctx->params.homopolymer_length; t.min_read_through_distance = ctx->params.min_read_through_distance; t.max_itd_length = ctx->params.max_itd_length;
	t.external_duplicate_marking = ctx->params.external_duplicate_marking; t.top_expressed_viral_verdict = nullptr; t.low_coverage_viral_verdict = nullptr;
}

extern "C" {

const char* emu_last_error(void) { return g_error.c_str(); }
int emu_api_version(void) { return AGPU_API_VERSION; }
int emu_device_count(void) { return 1; }
void emu_default_params(agpu_params* p) {
	memset(p, 0, sizeof(*p));
	p->homopolymer_length = 6; p->min_read_through_distance = 10000; p->max_itd_length = 100; p->subsampling_threshold = 300;
	p->mismatch_pvalue_cutoff = 0.01; p->max_kmer_content = 0.6; p->evalue_cutoff = 0.3; p->max_mismapper_fraction = 0.8; p->fragment_length = 200; p->exonic_fraction = 0.33; p->min_support = 2;
	for (int f = 1; f < AGPU_FILTER_COUNT; ++f) p->filter_enabled[f] = 1;
}
// (agpu_debug_fail_allocation_in_finish on the harness: the next `count` calls of emu_ingest_finish behave as the device does when an allocation fails inside agpu_ingest_finish --
// a context that is alone gives back its idle buffers and goes on; the lanes of a session, whose contexts share a pool, fail the sample with the device library's message)
static int g_live_contexts = 0, g_failing_allocations = 0;
void emu_debug_fail_allocation_in_finish(int count) { g_failing_allocations = count; }
void emu_debug_exhaust_memory_in_finish(int finishes) { g_failing_allocations = finishes; } // (the harness's hook has that meaning already)
emu_ctx* emu_create(int, const agpu_params* params) { emu_ctx* ctx = new emu_ctx(); if (params) ctx->params = *params; else emu_default_params(&ctx->params); memset(ctx->stage_counts, 0, sizeof(ctx->stage_counts)); ++g_live_contexts; return ctx; }
emu_ctx* emu_create_sibling(emu_ctx* of) { return of ? emu_create(0, &of->params) : nullptr; } // (the harness has no scratch buffers to share)
int emu_keep_batch_buffers(emu_ctx*, int) { return AGPU_OK; } // (... and none to hand from lane to lane: every context of the harness owns its batch)
void emu_destroy(emu_ctx* ctx) { if (ctx) --g_live_contexts; delete ctx; }
// (RCCL is the device library's: a workflow driver built over the harness is given a communicator of host collectives -- arriba_workflow_set_communicator -- and never gets here)
static int no_rccl() { g_error = "the stepping harness has no RCCL: hand the session a communicator of host collectives (arriba_workflow_set_communicator)"; return AGPU_ERR_INVALID; }
int emu_shard_merge_rccl(emu_ctx*, void*, uint32_t, agpu_ingest_result*) { return no_rccl(); }
int emu_filter_mismappers_rccl(emu_ctx*, void*, int32_t, uint32_t, uint32_t, uint64_t*, uint64_t*) { return no_rccl(); }
int emu_rccl_unique_id(uint8_t*) { return no_rccl(); }
int emu_rccl_join(emu_ctx*, const uint8_t*, uint32_t, uint32_t, void**) { return no_rccl(); }
int emu_rccl_leave(void*) { return AGPU_OK; }
int emu_rccl_all_gather_host(emu_ctx*, void*, uint32_t, const void*, void*, uint64_t) { return no_rccl(); }
int emu_rccl_all_reduce_host(emu_ctx*, void*, void*, uint64_t, int) { return no_rccl(); }
int emu_rccl_all_gather_device(emu_ctx*, void*, const void*, void*, uint64_t) { return no_rccl(); }
int emu_rccl_all_reduce_device(emu_ctx*, void*, void*, uint64_t, int) { return no_rccl(); }
int emu_scratch_buffer(emu_ctx*, const char* name, uint64_t bytes, void** pointer) { // (the harness has no device memory: host memory by name)
	static std::map<std::string, std::vector<uint8_t> > buffers; static std::mutex mutex;
	std::lock_guard<std::mutex> lock(mutex);
	std::vector<uint8_t>& buffer = buffers[name];
	if (buffer.size() < std::max<uint64_t>(bytes, 16)) buffer.resize(std::max<uint64_t>(bytes, 16));
	*pointer = buffer.data();
	return AGPU_OK;
}
int emu_device_copy(emu_ctx*, void* destination, const void* source, uint64_t bytes) { if (bytes > 0) memmove(destination, source, bytes); return AGPU_OK; }
int emu_set_params(emu_ctx* ctx, const agpu_params* params) { ctx->params = *params; if (ctx->n) build_tables(ctx); return 0; }

int emu_upload_annotation(emu_ctx* ctx, const agpu_annotation_view* in) {
	ctx->in_annotation = *in;
	ctx->n_real_genes = in->n_genes;
	ctx->gene_contig.assign(in->gene_contig, in->gene_contig + in->n_genes); ctx->gene_start.assign(in->gene_start, in->gene_start + in->n_genes); ctx->gene_end.assign(in->gene_end, in->gene_end + in->n_genes);
	ctx->gene_bits.assign(in->gene_bits, in->gene_bits + in->n_genes); ctx->gene_exonic_length.assign(in->gene_exonic_length, in->gene_exonic_length + in->n_genes);
	ctx->dummy_start_key.clear(); ctx->dummy_end_key.clear();
	refresh_annotation(ctx);
	return 0;
}
int emu_upload_genome(emu_ctx* ctx, const agpu_genome_view* in) {
	ctx->in_genome = *in;
	ctx->genome.n_contigs = in->n_contigs; ctx->genome.contig_offset = in->contig_offset; ctx->genome.contig_bits = in->contig_bits; ctx->genome.bases = in->bases;
	ctx->genome_size = 0;
	for (uint32_t c = 0; c < in->n_contigs; ++c) if (in->contig_bits[c] & AGPU_CBIT_INTERESTING) ctx->genome_size += in->contig_offset[c + 1] - in->contig_offset[c];
	if (ctx->n) build_tables(ctx);
	return 0;
}
int emu_upload_batch(emu_ctx* ctx, const agpu_batch_view* in) {
	const uint64_t n = in->n;
	ctx->n = n;
	ctx->fbits.assign(in->fbits, in->fbits + n); ctx->filter.assign(n, 0);
	ctx->read_sharded = false; ctx->state_imported = false; ctx->global_n = 0; ctx->have_sample_gene_read_counts = false;
	BatchView& b = ctx->batch;
	b.n = n; b.first_rank = 0; b.n_aln = in->n_aln; b.fbits = ctx->fbits.data(); b.filter = ctx->filter.data(); b.group = in->group;
	for (int s = 0; s < 3; ++s) {
		ctx->abits[s].assign(in->abits[s], in->abits[s] + n); ctx->gene_count[s].assign(n, 0); ctx->genes[s].assign(n * GENE_INLINE, 0);
		b.contig[s] = in->contig[s]; b.start[s] = in->start[s]; b.end[s] = in->end[s]; b.abits[s] = ctx->abits[s].data();
		b.cigar_offset[s] = in->cigar_offset[s]; b.cigar_count[s] = in->cigar_count[s]; b.gene_count[s] = ctx->gene_count[s].data(); b.genes[s] = ctx->genes[s].data();
	}
	b.cigar_pool = in->cigar_pool;
	ctx->max_read_length = 0;
	for (int s = 0; s < 2; ++s) { b.seq_offset[s] = in->seq_offset[s]; b.seq_length[s] = in->seq_length[s]; for (uint64_t i = 0; i < n; ++i) ctx->max_read_length = std::max(ctx->max_read_length, in->seq_length[s][i]); }
	b.seq_pool = in->seq_pool;
	ctx->gene_pool.assign(n / 2 + (1u << 20), 0); ctx->gene_pool_used = 0;
	b.gene_pool = ctx->gene_pool.data(); b.gene_pool_used = &ctx->gene_pool_used; b.gene_pool_capacity = ctx->gene_pool.size();
	memset(ctx->stage_counts, 0, sizeof(ctx->stage_counts));
	build_tables(ctx);
	return 0;
}

int emu_reset(emu_ctx*) { g_error = "emu_reset is not provided by the stepping harness"; return AGPU_ERR_INVALID; }

int emu_mark_multimappers(emu_ctx* ctx, uint64_t* marked) {
	BatchView& b = ctx->batch;
	uint64_t count = 0;
	for (uint64_t i = 0; i < b.n; ++i) {
		bool previous = i > 0 && b.group[i - 1] == b.group[i], next = i + 1 < b.n && b.group[i + 1] == b.group[i];
		if (previous || next) b.fbits[i] |= FBIT_MULTIMAPPER;
		if (next) ++count;
	}
	if (marked) *marked = count;
	return 0;
}

int emu_set_shard(emu_ctx* ctx, uint64_t first_rank, uint64_t global_n) { ctx->batch.first_rank = first_rank; ctx->global_n = global_n; return 0; }

int emu_annotate_begin(emu_ctx* ctx, uint64_t* n_unmapped) {
	BatchView& b = ctx->batch;
	ctx->unmapped.assign(2 * b.n + 2, 0);
	uint32_t unmapped_count = 0;
	bool ok = true;
	for (uint64_t i = 0; i < b.n; ++i)
	{
		uint64_t positions[2]; uint32_t n_positions;
		ok = annotate_fragment_stage1(b, ctx->annotation, ctx->params.strandedness, i, positions, n_positions) && ok;
		for (uint32_t k = 0; k < n_positions; ++k) ctx->unmapped[unmapped_count++] = positions[k];
	}
	if (!ok) { g_error = "a gene set exceeded the device capacity"; return AGPU_ERR_CAPACITY; }
	ctx->unmapped.resize(unmapped_count);
	if (n_unmapped) *n_unmapped = unmapped_count;
	return 0;
}
int emu_copy_unmapped_positions(emu_ctx* ctx, uint64_t* destination) { memcpy(destination, ctx->unmapped.data(), ctx->unmapped.size() * 8); return 0; }

int emu_annotate_finish(emu_ctx* ctx, const uint64_t* positions, uint64_t n_positions, uint32_t* n_dummy_genes) {
	BatchView& b = ctx->batch;
	bool ok = true;
	std::vector<uint64_t> unmapped = positions ? std::vector<uint64_t>(positions, positions + n_positions) : ctx->unmapped;
	const uint32_t unmapped_count = unmapped.size();
	std::sort(unmapped.begin(), unmapped.end());
	ctx->dummy_start_key.clear(); ctx->dummy_end_key.clear();
	for (uint32_t i = 0; i < unmapped_count; ++i) {
		if (dummy_gene_starts_here(unmapped.data(), i, ctx->annotation.gene_index)) { ctx->dummy_start_key.push_back(unmapped[i]); ctx->dummy_end_key.push_back(unmapped[i]); }
		else ctx->dummy_end_key.back() = unmapped[i];
	}
	const uint32_t n_dummy = ctx->dummy_start_key.size();
	ctx->gene_contig.resize(ctx->n_real_genes + n_dummy); ctx->gene_start.resize(ctx->n_real_genes + n_dummy); ctx->gene_end.resize(ctx->n_real_genes + n_dummy);
	ctx->gene_bits.resize(ctx->n_real_genes + n_dummy); ctx->gene_exonic_length.resize(ctx->n_real_genes + n_dummy);
	for (uint32_t j = 0; j < n_dummy; ++j) {
		uint32_t g = ctx->n_real_genes + j;
		ctx->gene_contig[g] = ctx->dummy_start_key[j] >> 32; ctx->gene_start[g] = (int32_t) (uint32_t) ctx->dummy_start_key[j]; ctx->gene_end[g] = (int32_t) (uint32_t) ctx->dummy_end_key[j];
		ctx->gene_bits[g] = GBIT_STRAND | GBIT_DUMMY; ctx->gene_exonic_length[g] = 10000;
	}
	refresh_annotation(ctx);
	ctx->viral_pairs.clear();
	for (uint64_t i = 0; i < b.n; ++i) {
		ok = annotate_fragment_stage2(b, ctx->annotation, i) && ok;
		int mate2 = (b.n_aln[i] == 3) ? SUPPLEMENTARY : MATE2;
		int viral_slot = -1, host_slot = -1;
		uint8_t bits1 = ctx->genome.contig_bits[b.contig[MATE1][i]], bits2 = ctx->genome.contig_bits[b.contig[mate2][i]];
		if (bits1 & CBIT_VIRAL) viral_slot = MATE1; else if (bits1 & CBIT_INTERESTING) host_slot = MATE1;
		if (bits2 & CBIT_VIRAL) viral_slot = mate2; else if (bits2 & CBIT_INTERESTING) host_slot = mate2;
		if (viral_slot >= 0 && host_slot >= 0) {
			AGPU_IDSET(genes); load_genes(b, host_slot, i, genes);
			for (uint32_t g = 0; g < genes.n; ++g) { ctx->viral_pairs.push_back(b.contig[viral_slot][i]); ctx->viral_pairs.push_back(genes.get(g)); }
		}
	}
	if (!ok) { g_error = "a gene set exceeded the device capacity"; return AGPU_ERR_CAPACITY; }
	if (n_dummy_genes) *n_dummy_genes = n_dummy;
	return 0;
}
int emu_annotate(emu_ctx* ctx, uint32_t* n_dummy_genes) {
	int status = emu_annotate_begin(ctx, nullptr);
	return status ? status : emu_annotate_finish(ctx, nullptr, 0, n_dummy_genes);
}

int emu_get_viral_integration_sites(emu_ctx* ctx, uint32_t* pairs, uint64_t capacity, uint64_t* count) {
	uint64_t available = ctx->viral_pairs.size() / 2;
	if (count) *count = available;
	if (pairs) memcpy(pairs, ctx->viral_pairs.data(), std::min(available, capacity) * 8);
	return 0;
}

struct EmuDuplicateEntry { DuplicateKey key; uint32_t rank; };
static int emu_stage1(emu_ctx* ctx, const uint8_t* top_verdict, const uint8_t* low_verdict, const EmuDuplicateEntry* entries, uint64_t n_entries, bool export_winners);
int emu_read_filters_stage1(emu_ctx* ctx, const uint8_t* top_verdict, const uint8_t* low_verdict) { return emu_stage1(ctx, top_verdict, low_verdict, nullptr, 0, false); }
int emu_duplicates_begin(emu_ctx* ctx, uint64_t* n_entries) {
	int status = emu_stage1(ctx, nullptr, nullptr, nullptr, 0, true);
	if (n_entries) *n_entries = ctx->duplicate_entries.size() / sizeof(EmuDuplicateEntry);
	return status;
}
int emu_copy_duplicate_entries(emu_ctx* ctx, void* destination) { memcpy(destination, ctx->duplicate_entries.data(), ctx->duplicate_entries.size()); return 0; }
int emu_read_filters_stage1_global(emu_ctx* ctx, const void* entries, uint64_t n_entries, const uint8_t* top_verdict, const uint8_t* low_verdict) {
	return emu_stage1(ctx, top_verdict, low_verdict, (const EmuDuplicateEntry*) entries, n_entries, false);
}

static int emu_stage1(emu_ctx* ctx, const uint8_t* top_verdict, const uint8_t* low_verdict, const EmuDuplicateEntry* entries, uint64_t n_entries, bool export_winners) {
	BatchView& b = ctx->batch;
	if (top_verdict) { ctx->verdict_top.assign(top_verdict, top_verdict + ctx->genome.n_contigs); ctx->tables.top_expressed_viral_verdict = ctx->verdict_top.data(); } else ctx->tables.top_expressed_viral_verdict = nullptr;
	if (low_verdict) { ctx->verdict_low.assign(low_verdict, low_verdict + ctx->genome.n_contigs); ctx->tables.low_coverage_viral_verdict = ctx->verdict_low.data(); } else ctx->tables.low_coverage_viral_verdict = nullptr;
	const uint8_t* enabled = ctx->params.filter_enabled;
	memset(ctx->stage_counts, 0, sizeof(ctx->stage_counts));
	// same open-addressing table as the device path, filled sequentially
	uint64_t slots = 1024;
	while (slots < 2 * b.n) slots <<= 1;
	uint32_t mask = slots - 1;
	std::vector<uint32_t> table(slots, 0xFFFFFFFFu);
	std::vector<DuplicateKey> keys(b.n);
	for (uint64_t i = 0; i < b.n; ++i) keys[i] = duplicate_key(b, i);
	for (uint64_t i = 0; i < b.n; ++i) {
		uint32_t h = (uint32_t) hash_duplicate_key(keys[i]) & mask;
		while (true) {
			if (table[h] == 0xFFFFFFFFu) { table[h] = i; break; }
			if (keys_equal(keys[table[h]], keys[i])) { table[h] = std::min<uint32_t>(table[h], i); break; }
			h = (h + 1) & mask;
		}
	}
	if (export_winners) { // the local winners in name order, for the exchange between shards; no filtering yet
		ctx->duplicate_entries.clear();
		for (uint64_t i = 0; i < b.n; ++i) {
			uint32_t h = (uint32_t) hash_duplicate_key(keys[i]) & mask;
			while (!keys_equal(keys[table[h]], keys[i])) h = (h + 1) & mask;
			if (table[h] != i) continue;
			EmuDuplicateEntry entry; entry.key = keys[i]; entry.rank = (uint32_t) (b.first_rank + i);
			const uint8_t* bytes = (const uint8_t*) &entry;
			ctx->duplicate_entries.insert(ctx->duplicate_entries.end(), bytes, bytes + sizeof(entry));
		}
		return 0;
	}
	std::map<std::tuple<uint32_t, int32_t, int32_t>, uint32_t> global_winner; // key -> smallest global name rank over all shards
	for (uint64_t e = 0; e < n_entries; ++e) {
		auto key = std::make_tuple(entries[e].key.contigs, entries[e].key.position1, entries[e].key.position2);
		auto found = global_winner.find(key);
		if (found == global_winner.end() || entries[e].rank < found->second) global_winner[key] = entries[e].rank;
	}
	for (uint64_t i = 0; i < b.n; ++i) {
		uint8_t filter = b.filter[i];
		if (filter == FILTER_none && enabled[FILTER_duplicates]) {
			if (ctx->tables.external_duplicate_marking) { if (b.fbits[i] & FBIT_DUPLICATE) filter = FILTER_duplicates; }
			else if (entries != nullptr) {
				if (global_winner.at(std::make_tuple(keys[i].contigs, keys[i].position1, keys[i].position2)) != (uint32_t) (b.first_rank + i)) filter = FILTER_duplicates;
			} else {
				uint32_t h = (uint32_t) hash_duplicate_key(keys[i]) & mask;
				while (!keys_equal(keys[table[h]], keys[i])) h = (h + 1) & mask;
				if (table[h] != i) filter = FILTER_duplicates;
			}
			if (filter != FILTER_none) ctx->stage_counts[0]++;
		}
		if (filter == FILTER_none) {
			uint8_t hit = contig_filters(b, ctx->genome, ctx->tables, i);
			if (hit != FILTER_none && enabled[hit]) {
				filter = hit;
				ctx->stage_counts[hit == FILTER_uninteresting_contigs ? 1 : hit == FILTER_viral_contigs ? 2 : hit == FILTER_top_expressed_viral_contigs ? 3 : 4]++;
			}
		}
		b.filter[i] = filter;
	}
	return 0;
}

int emu_fragment_length_samples_limited(emu_ctx* ctx, uint32_t limit, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited);
int emu_fragment_length_samples(emu_ctx* ctx, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited) { return emu_fragment_length_samples_limited(ctx, MAX_SAMPLES, mate_gaps, n_samples, fragments_visited); }
int emu_fragment_length_samples_limited(emu_ctx* ctx, uint32_t limit, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited) {
	BatchView& b = ctx->batch;
	uint32_t count = 0;
	uint64_t visited = b.n;
	for (uint64_t i = 0; i < b.n; ++i) {
		if (b.filter[i] == FILTER_none && !(b.fbits[i] & FBIT_SINGLE_END) && b.n_aln[i] == 3) {
			if (mate_gaps) mate_gaps[count] = mate_gap_sample(b, ctx->annotation, i);
			++count;
			if (count == limit) { visited = i + 1; break; }
		}
	}
	if (n_samples) *n_samples = count;
	if (fragments_visited) *fragments_visited = visited;
	return 0;
}

int emu_read_filters_stage2(emu_ctx* ctx, uint64_t* remaining) {
	BatchView& b = ctx->batch;
	for (uint64_t i = 0; i < b.n; ++i) {
		uint32_t first_hit;
		b.filter[i] = read_filters_stage2(b, ctx->annotation, ctx->genome, ctx->tables, ctx->params.filter_enabled, i, b.filter[i], no_stage(), first_hit);
		if (first_hit < 9) ctx->stage_counts[5 + first_hit]++;
	}
	if (ctx->params.filter_enabled[FILTER_low_entropy])
		for (uint64_t i = 0; i < b.n; ++i) {
			uint8_t filter = b.filter[i];
			if (needs_low_entropy_test(b, ctx->tables, i, filter) && has_low_entropy(b, ctx->tables, i, no_stage())) {
				if (filter == FILTER_none) ctx->stage_counts[13]++;
				b.filter[i] = FILTER_low_entropy;
			}
		}
	if (remaining) {
		static const int order[14] = { 1, 30, 31, 32, 33, 4, 2, 3, 6, 7, 5, 8, 10, 36 };
		for (int f = 0; f < AGPU_FILTER_COUNT; ++f) remaining[f] = 0;
		uint64_t left = b.n;
		for (int k = 0; k < 14; ++k) { left -= ctx->stage_counts[k]; remaining[order[k]] = left; }
	}
	return 0;
}

int emu_get_filters(emu_ctx* ctx, uint8_t* filter) {
	if (ctx->read_sharded) { if (!ctx->state_imported) { g_error = "agpu_read_state_import must run first"; return AGPU_ERR_INVALID; } memcpy(filter, ctx->global_filter.data(), ctx->global_n); return 0; } // (the fragments of the sample)
	memcpy(filter, ctx->filter.data(), ctx->n); return 0;
}
int emu_get_filters_of(emu_ctx* ctx, const uint32_t* fragments, uint64_t n, uint8_t* filter) {
	if (ctx->read_sharded) { // (global name ranks)
		if (!ctx->state_imported) { g_error = "agpu_read_state_import must run first"; return AGPU_ERR_INVALID; }
		for (uint64_t k = 0; k < n; ++k) { if (fragments[k] >= ctx->global_n) { g_error = "fragment index out of range"; return AGPU_ERR_INVALID; } filter[k] = ctx->global_filter[fragments[k]]; }
		return 0;
	}
	for (uint64_t k = 0; k < n; ++k) { if (fragments[k] >= ctx->n) { g_error = "fragment index out of range"; return AGPU_ERR_INVALID; } filter[k] = ctx->filter[fragments[k]]; }
	return 0;
}
int emu_get_alignment_bits(emu_ctx* ctx, int slot, uint8_t* abits) { memcpy(abits, ctx->abits[slot].data(), ctx->n); return 0; }
int emu_get_fragment_bits(emu_ctx* ctx, uint8_t* fbits) { memcpy(fbits, ctx->fbits.data(), ctx->n); return 0; }
int emu_get_gene_sets(emu_ctx* ctx, int slot, uint8_t* count, uint32_t* genes, uint64_t capacity, uint64_t* total) {
	uint64_t sum = 0;
	for (uint64_t i = 0; i < ctx->n; ++i) sum += ctx->gene_count[slot][i];
	if (total) *total = sum;
	if (count) memcpy(count, ctx->gene_count[slot].data(), ctx->n);
	if (genes) {
		if (capacity < sum) { g_error = "gene buffer too small"; return AGPU_ERR_INVALID; }
		uint64_t at = 0;
		for (uint64_t i = 0; i < ctx->n; ++i) {
			AGPU_IDSET(set); load_genes(ctx->batch, slot, i, set);
			for (uint32_t k = 0; k < set.n; ++k) genes[at++] = set.get(k);
		}
	}
	return 0;
}
int emu_get_gene_table(emu_ctx* ctx, uint32_t first, uint32_t count, uint16_t* contig, int32_t* start, int32_t* end, uint8_t* bits, int32_t* exonic_length) {
	if ((size_t) first + count > ctx->gene_contig.size()) { g_error = "gene range out of bounds"; return AGPU_ERR_INVALID; }
	if (contig) memcpy(contig, ctx->gene_contig.data() + first, count * 2);
	if (start) memcpy(start, ctx->gene_start.data() + first, count * 4);
	if (end) memcpy(end, ctx->gene_end.data() + first, count * 4);
	if (bits) memcpy(bits, ctx->gene_bits.data() + first, count);
	if (exonic_length) memcpy(exonic_length, ctx->gene_exonic_length.data() + first, count * 4);
	return 0;
}
int emu_last_kernel_ms(emu_ctx*, float* ms) { *ms = 0; return 0; }
int emu_last_kernel_bytes(emu_ctx*, uint64_t* bytes) { *bytes = 0; return 0; }
// (there are no launches to time here: one placeholder sample, so that callers that read a profile -- bench.py under tests/bench_on_harness.py -- find its shape)
int emu_set_profiling(emu_ctx*, int) { return 0; }
int emu_get_kernel_profile(emu_ctx*, char* names, float* ms, uint64_t* bytes, uint32_t capacity, uint32_t* count) {
	if (count) *count = 1;
	if (capacity >= 1 && names && ms && bytes) { memset(names, 0, AGPU_KERNEL_NAME_LENGTH); strcpy(names, "host_stepping_harness"); ms[0] = 1.0f; bytes[0] = 1; }
	return 0;
}

#include "emu_fusions.inc"
#include "emu_ingest.inc"
#include "emu_sharded.inc"

}
