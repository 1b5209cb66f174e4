// arriba_amd/csrc/workflow/workflow.cpp -- the order in which the reference's main() calls its stages (source/arriba.cpp:84-615), over the two C ABIs.
// Host C++ like the reference; no arithmetic of the path lives here: every step is a call into libarriba_host (loaders, ingest, sequential
// scalar stages, writer) or libarriba_gpu (the stages on the device).  arriba_amd/pipeline.py: DevicePipeline.run_workflow is the same sequence for
// the tests and the bench.
#include "../../../include/arriba_workflow.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <functional>
#include <memory>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

std::string g_error;
int g_error_code = 0; // the status of the device library behind g_error (AGPU_ERR_*), 0 = the failure was not the device's

struct Failure { std::string text; int code = 0; Failure(const std::string& t, int c = 0) : text(t), code(c) {} };
void device_check(int status) { if (status != AGPU_OK) throw Failure{ std::string("ERROR: ") + agpu_last_error(), status }; }
void host_check(int status) { if (status != 0) throw Failure{ std::string("ERROR: ") + ahost_last_error() }; }

// filter ids (positions in FILTERS, source/common.hpp:29-67) of the filters main() asks about itself
enum { F_known_fusions = 18, F_blacklist = 20, F_no_genomic_support = 29, F_genomic_support = 34, F_many_spliced = 28, F_select_best = 24 };

double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One sample over several ranks: in front of every collective the ranks tell each other how they fared, so that an error on one of them (a damaged block in its part of the
// file, no memory) ends the call on all of them with a message instead of leaving the others waiting in an all-gather (arriba_amd/one_sample.py: _together)
void exchange_check(int status, const char* what) { if (status != 0) throw Failure{ std::string("ERROR: the exchange between the ranks failed (") + what + ")" }; }
template <class Work> void together(const arriba_workflow_communicator* ranks, Work work) {
	if (ranks == nullptr) { work(); return; }
	std::string problem; int code = 0;
	try { work(); }
	catch (const Failure& failure) { problem = failure.text; code = failure.code; }
	catch (const std::exception& e) { problem = std::string("ERROR: ") + e.what(); }
	int64_t ok = problem.empty() ? 1 : 0;
	const int status = ranks->all_reduce_int64(ranks->state, &ok, 1, ARRIBA_WORKFLOW_MIN);
	if (!problem.empty()) throw Failure{ problem, code };
	exchange_check(status, "status of the ranks");
	if (ok == 0) throw Failure{ "ERROR: another rank of the sample failed (its own message says why)" };
}

// A session (arriba_workflow_open): what is loaded once, and what a sample leaves behind for the next one.  `options` points at copies of the caller's strings.
struct Run {
	arriba_workflow_options options;
	std::vector<std::string> strings; // the texts options.* point at (the caller's may be gone after arriba_workflow_open)
	arriba_workflow_report* report;
	arriba_workflow_timing* timing;
	ahost_session* host;
	agpu_ctx* device;
	uint32_t dummy_genes;
	uint64_t n_candidates, n_fragments, mapped_reads;
	bool device_ingest;
	enum { FEED_BUFFERS = 4 }; // (see read_chimeric_alignments_on_device)
	void* pieces[FEED_BUFFERS];
	size_t piece_bytes = 0;
	agpu_bgzf_block* tables[FEED_BUFFERS]; // pinned like the pieces: a copy from pageable memory is staged by the runtime when the stream gets there, and the caller waits for that
	agpu_params params; // as the last sample resolved them (strandedness)
	// read_chimeric_alignments in two halves (feed_file: the bytes of the file into HBM, on a thread of its own when the sample was submitted ahead; finish_device_ingest: what is
	// left behind the last piece): what the first half leaves for the second
	bool bam_open = false; uint64_t coverage_windows = 0; uint32_t bam_contigs = 0; double feed_started = 0, feed_finished = 0, feed_reading = 0, feed_pushing = 0;
	std::string bam_path, output_path, discarded_path; // of the sample this lane works on (options.* point at them)
	std::function<void()> after_ingest; // a session with two lanes: the stream and the tables of the ingest are free for the feed of the next sample
	// arriba_workflow_finish_ahead: the second half ran on the feeder thread, before the caller asked for the sample; what it found is noted (report, timing) when the caller does
	bool ingest_finished_ahead = false; uint64_t ingest_records = 0, ingest_stream_bytes = 0; double ingest_fed = 0, ingest_finished = 0, ingest_adopted = 0;
	// arriba_workflow_defer_output: the last file of a sample is formatted and written by a thread of its own once everything it needs has left the device, while the caller goes on
	// with the next sample; what that thread reads stays with the lane until it is joined (the staged columns and rows, the gene table below)
	bool defer_output = false;
	std::function<void()> before_host_writer; // (the formatter threads of the host library serve one writer at a time: the writer of the other lane is waited for)
	std::thread writer; std::string writer_error; double writer_seconds = 0;
	std::vector<uint16_t> writer_gene_contig; std::vector<int32_t> writer_gene_start, writer_gene_end;
	void join_writer() { if (writer.joinable()) writer.join(); }
	bool tags_loaded = false, domains_loaded = false;
	// one sample over the ranks of a job (arriba_workflow_set_communicator): the session's communicator, or null; what the exchanges of the sample at work took
	const arriba_workflow_communicator* ranks = nullptr;
	double exchange_parts = 0, exchange_verdicts = 0, exchange_rows = 0;
	// ... the reads of the sample at work sharded over the ranks (shard_reads below): this rank holds the fragments [first_rank, first_rank + local_fragments) of n_fragments
	bool sharded = false; uint64_t first_rank = 0, local_fragments = 0;
	uint64_t exchanged_bytes = 0; // what this rank received in the exchanges of the sample at work (the part that is not its own)
	std::vector<uint8_t> exchange_mine, exchange_all; // host buffers of gather_from_device (kept: a gigabyte of fresh vector per exchange is zeroed page by page)
	// Host memory for what comes back from the device per sample (candidate columns, rows of supporting reads): pinned, kept by the session and only ever grown -- fresh
	// std::vectors of some hundred MB per sample are zeroed page by page and given back to the system again, which costs more than the transfer they hold.
	struct Staged { void* pointer = nullptr; size_t capacity = 0; };
	std::map<std::string, Staged> staged;
	template <class T> T* stage(const char* name, size_t count) {
		Staged& buffer = staged[name];
		const size_t bytes = (count > 0 ? count : 1) * sizeof(T);
		if (bytes > buffer.capacity) {
			if (buffer.pointer) agpu_host_free(buffer.pointer);
			buffer.capacity = bytes + bytes / 4;
			buffer.pointer = agpu_host_alloc(buffer.capacity);
			if (!buffer.pointer) { buffer.capacity = 0; throw Failure{ std::string("ERROR: ") + agpu_last_error() }; }
		}
		return (T*) buffer.pointer;
	}
	Run(const arriba_workflow_options& o): options(o), report(nullptr), timing(nullptr), host(nullptr), device(nullptr), dummy_genes(0), n_candidates(0), n_fragments(0), mapped_reads(0), device_ingest(false) {
		for (int k = 0; k < FEED_BUFFERS; ++k) { pieces[k] = nullptr; tables[k] = nullptr; }
		const char** texts[] = { &options.assembly_file, &options.gene_annotation_file, &options.chimeric_bam_file, &options.output_file, &options.discarded_output_file, &options.blacklist_file, &options.known_fusions_file,
		                         &options.tags_file, &options.protein_domains_file, &options.genomic_breakpoints_file, &options.interesting_contigs, &options.viral_contigs, &options.gtf_features };
		strings.reserve(sizeof(texts) / sizeof(texts[0]));
		for (size_t k = 0; k < sizeof(texts) / sizeof(texts[0]); ++k) if (*texts[k] != nullptr) { strings.push_back(*texts[k]); *texts[k] = strings.back().c_str(); }
	}
	~Run() {
		join_writer();
		for (int k = 0; k < FEED_BUFFERS; ++k) { if (pieces[k]) agpu_host_free(pieces[k]); if (tables[k]) agpu_host_free(tables[k]); }
		for (std::map<std::string, Staged>::iterator buffer = staged.begin(); buffer != staged.end(); ++buffer) if (buffer->second.pointer) agpu_host_free(buffer->second.pointer);
		if (device) agpu_destroy(device); if (host) ahost_close(host);
	}
	Run(const Run&) = delete; Run& operator=(const Run&) = delete;
	void note(const char* stage, uint64_t count) {
		progress(stage, count);
		if (!report || report->n_stages >= sizeof(report->stages) / sizeof(report->stages[0])) return;
		arriba_workflow_stage& entry = report->stages[report->n_stages++];
		snprintf(entry.stage, sizeof(entry.stage), "%s", stage);
		entry.count = count;
	}
	// the progress lines of the reference's main() (source/arriba.cpp:96-610), one per stage; a filter switched off with -f prints none, as there
	static std::string time_string() { time_t now = time(0); char buffer[100]; strftime(buffer, sizeof(buffer), "[%Y-%m-%dT%X]", localtime(&now)); return buffer; }
	void say(const std::string& text) const { if (options.log_to_stdout) std::cout << time_string() << " " << text << std::endl; }
	void progress(const std::string& stage, uint64_t count) const {
		if (!options.log_to_stdout) return;
		const arriba_workflow_options& o = options;
		std::ostringstream text; // (stream formatting of the numbers, as std::cout formats them in the reference)
		unsigned filter = 0; const char* what = "remaining";
		if (stage == "mark_multimappers") { text << "Marking multi-mapping alignments "; what = "marked"; }
		else if (stage == "filter_duplicates") { filter = 1; text << "Filtering duplicates "; }
		else if (stage == "filter_uninteresting_contigs") { filter = 30; text << "Filtering mates which do not map to interesting contigs (" << (o.interesting_contigs ? o.interesting_contigs : "1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 X Y AC_* NC_*") << ") "; }
		else if (stage == "filter_viral_contigs") { filter = 31; text << "Filtering mates which only map to viral contigs (" << (o.viral_contigs ? o.viral_contigs : "AC_* NC_*") << ") "; }
		else if (stage == "filter_top_expressed_viral_contigs") { filter = 32; text << "Filtering viral contigs with expression lower than the top " << o.top_viral_contigs << " "; }
		else if (stage == "filter_low_coverage_viral_contigs") { filter = 33; text << "Filtering viral contigs with less than " << (o.viral_contig_min_covered_fraction * 100) << "% coverage "; }
		else if (stage == "filter_proximal_read_through") { filter = 4; text << "Filtering read-through fragments with a distance <=" << o.device.min_read_through_distance << "bp "; }
		else if (stage == "filter_inconsistently_clipped_mates") { filter = 2; text << "Filtering inconsistently clipped mates "; }
		else if (stage == "filter_homopolymer") { filter = 3; text << "Filtering breakpoints adjacent to homopolymers >=" << o.device.homopolymer_length << "nt "; }
		else if (stage == "filter_small_insert_size") { filter = 6; text << "Filtering fragments with small insert size "; }
		else if (stage == "filter_long_gap") { filter = 7; text << "Filtering alignments with long gaps "; }
		else if (stage == "filter_same_gene") { filter = 5; text << "Filtering fragments with both mates in the same gene "; }
		else if (stage == "filter_hairpin") { filter = 8; text << "Filtering fusions arising from hairpin structures "; }
		else if (stage == "filter_mismatches") { filter = 10; text << "Filtering reads with a mismatch p-value <=" << o.device.mismatch_pvalue_cutoff << " "; }
		else if (stage == "filter_low_entropy") { filter = 36; text << "Filtering reads with low entropy (k-mer content >=" << (o.device.max_kmer_content * 100) << "%) "; }
		else if (stage == "find_fusions") { text << "Finding fusions and counting supporting reads "; what = "total"; }
		else if (stage == "mark_genomic_support") { text << "Marking fusions with support from whole-genome sequencing in '" << o.genomic_breakpoints_file << "' "; what = "marked"; }
		else if (stage == "merge_adjacent_fusions") { filter = 23; text << "Merging adjacent fusion breakpoints "; }
		else if (stage == "filter_multimappers") { filter = 9; text << "Filtering multi-mapping fusions by alignment score and read support "; }
		else if (stage == "filter_non_coding_neighbors") { filter = 14; text << "Filtering fusions with both breakpoints in adjacent non-coding/intergenic regions "; }
		else if (stage == "filter_intragenic_both_exonic") { filter = 15; text << "Filtering intragenic fusions with both breakpoints in exonic regions "; }
		else if (stage == "filter_min_support") { filter = 17; text << "Filtering fusions with <" << o.device.min_support << " supporting reads "; }
		else if (stage == "filter_relative_support") { filter = 12; text << "Filtering fusions with an e-value >=" << o.device.evalue_cutoff << " "; }
		else if (stage == "recover_internal_tandem_duplication") { filter = 16; text << "Searching for internal tandem duplications <=" << o.device.max_itd_length << "bp with >=" << o.min_itd_support << " supporting reads and >=" << (o.min_itd_allele_fraction * 100) << "% allele fraction "; }
		else if (stage == "filter_both_intronic") { filter = 13; text << "Filtering fusions with both breakpoints in intronic/intergenic regions "; }
		else if (stage == "recover_known_fusions") { filter = 18; text << "Searching for known fusions in '" << o.known_fusions_file << "' "; }
		else if (stage == "filter_in_vitro") { filter = 22; text << "Filtering in vitro-generated fusions between genes with an expression above the " << (o.high_expression_quantile * 100) << "% quantile "; }
		else if (stage == "recover_both_spliced") { filter = 19; text << "Searching for fusions with spliced split reads "; }
		else if (stage == "select_most_supported_breakpoints") { filter = 24; text << "Selecting best breakpoints from genes with multiple breakpoints "; }
		else if (stage == "filter_marginal_read_through") { filter = 25; text << "Filtering read-through fusions with breakpoints near the gene boundary "; }
		else if (stage == "recover_many_spliced") { filter = 28; text << "Searching for fusions with >=" << o.min_spliced_events << " spliced events "; }
		else if (stage == "filter_no_genomic_support") { filter = 29; text << "Filtering low-confidence events with no support from WGS "; }
		else if (stage == "filter_blacklisted_ranges") { filter = 20; text << "Filtering blacklisted fusions in '" << o.blacklist_file << "' "; }
		else if (stage == "filter_short_anchor") { filter = 26; text << "Filtering fusions with anchors <=" << o.min_anchor_length << "nt "; }
		else if (stage == "filter_end_to_end_fusions") { filter = 21; text << "Filtering end-to-end fusions with low support "; }
		else if (stage == "filter_no_coverage") { filter = 27; text << "Filtering fusions with no coverage around the breakpoints "; }
		else if (stage == "filter_homologs") { filter = 37; text << "Filtering genes with >=" << (o.max_homolog_identity * 100) << "% identity "; }
		else if (stage == "filter_mismappers") { filter = 11; text << "Re-aligning chimeric reads to filter fusions with >=" << (o.device.max_mismapper_fraction * 100) << "% mis-mappers "; }
		else if (stage == "recover_genomic_support") { filter = 34; text << "Searching for fusions with support from WGS "; }
		else if (stage == "recover_isoforms") { filter = 35; text << "Searching for additional isoforms "; }
		else return;
		if (filter != 0 && !enabled(filter)) return;
		std::cout << time_string() << " " << text.str() << "(" << what << "=" << count << ")" << std::endl;
	}
	bool enabled(unsigned filter) const { return options.device.filter_enabled[filter] != 0; }
};

// read_chimeric_alignments on the device (source/read_chimeric_alignments.cpp:560-773): the host opens the file, parses the BAM header and feeds the bytes in
// pieces through two pinned buffers in turn; the batch and coverage_t are built in HBM, the host takes over the counters and coverage_t
void feed_file(Run& run) {
	const arriba_workflow_options& o = run.options;
	const double started = now_seconds();
	run.feed_started = started;
	agpu_ingest_config config;
	if (run.ranks != nullptr) host_check(ahost_bam_open_part(run.host, o.chimeric_bam_file, o.device.external_duplicate_marking, o.device.max_itd_length, run.ranks->rank, run.ranks->size, &config)); // this rank's part of the records
	else host_check(ahost_bam_open(run.host, o.chimeric_bam_file, o.device.external_duplicate_marking, o.device.max_itd_length, &config));
	run.bam_open = true;
	struct Closer { Run& run; bool armed; ~Closer() { if (armed && run.bam_open) { ahost_bam_close(run.host); run.bam_open = false; } } } closer = { run, true }; // (on every way out but the last line)
	run.coverage_windows = config.n_contigs ? config.coverage_window_offset[config.n_contigs] : 0; // (the table belongs to the session: read before anything else touches it)
	run.bam_contigs = config.n_contigs;
	device_check(agpu_upload_genome(run.device, ahost_genome_view(run.host))); // with the contigs of the BAM header
	// The file is read by one thread (ahost_bam_next: all cores pread into the next pinned buffer) while this one pushes the piece before (agpu_ingest_push*: enqueue the copy,
	// move the windows of the ingest on -- ~1 ms of runtime calls per piece, 0.17-0.26 s of a 54 GB file when the reader had to wait for them, profiles/r03o, r03p).  A push returns
	// when the piece pushed two pushes ago has left its buffer (host_buffers = 3); with four buffers in turn the buffer of piece k is free when push k - 2 has returned, so piece k
	// is read while piece k - 1 is pushed and neither waits for the other's bookkeeping.
	const unsigned int buffers = Run::FEED_BUFFERS;
	config.host_buffers = 3;
	device_check(agpu_ingest_begin(run.device, &config));
	size_t piece_bytes = 256u << 20;
	if (const char* knob = getenv("ARRIBA_FEED_PIECE_MB")) if (atoi(knob) >= 1 && atoi(knob) <= 256) piece_bytes = (size_t) atoi(knob) << 20; // (tests: many pieces of a small file through the reader and the pusher)
	if (run.piece_bytes != 0 && run.piece_bytes != piece_bytes) for (unsigned int k = 0; k < buffers; ++k) { if (run.pieces[k]) agpu_host_free(run.pieces[k]); if (run.tables[k]) agpu_host_free(run.tables[k]); run.pieces[k] = nullptr; run.tables[k] = nullptr; }
	run.piece_bytes = piece_bytes;
	const uint32_t block_capacity = (uint32_t) (piece_bytes / 4096 + 16);
	for (unsigned int k = 0; k < buffers; ++k) { // the pinned buffers stay with the session: pinning 2 x 256 MB costs as much as feeding a gigabyte
		if (!run.tables[k]) { run.tables[k] = (agpu_bgzf_block*) agpu_host_alloc((size_t) block_capacity * sizeof(agpu_bgzf_block)); if (!run.tables[k]) throw Failure{ std::string("ERROR: ") + agpu_last_error() }; }
		if (!run.pieces[k]) { run.pieces[k] = agpu_host_alloc(piece_bytes); if (!run.pieces[k]) throw Failure{ std::string("ERROR: ") + agpu_last_error() }; }
	}
	struct Feed {
		std::mutex mutex; std::condition_variable changed;
		unsigned int read = 0, pushed = 0; // pieces read / pushes that have returned
		bool at_end = false, stop = false; std::string error;
		ahost_bam_piece pieces[Run::FEED_BUFFERS];
		double reading = 0;
	} feed;
	std::thread reader([&run, &feed, buffers, piece_bytes, block_capacity] {
		for (unsigned int k = 0; ; ++k) {
			{ std::unique_lock<std::mutex> lock(feed.mutex); feed.changed.wait(lock, [&] { return feed.stop || k < buffers || feed.pushed + 1 >= k; }); if (feed.stop) return; } // (push k - 2 has returned: piece k - 4 has left this buffer)
			const double before = now_seconds();
			ahost_bam_piece piece;
			const int status = ahost_bam_next(run.host, run.pieces[k % buffers], piece_bytes, run.tables[k % buffers], block_capacity, &piece);
			std::lock_guard<std::mutex> lock(feed.mutex);
			feed.reading += now_seconds() - before;
			if (status < 0) feed.error = std::string("ERROR: ") + ahost_last_error(); // (the text belongs to this thread)
			if (status <= 0) { feed.at_end = true; feed.changed.notify_all(); return; }
			feed.pieces[k % buffers] = piece; feed.read = k + 1;
			feed.changed.notify_all();
		}
	});
	struct Joiner { std::thread& thread; Feed& feed; ~Joiner() { { std::lock_guard<std::mutex> lock(feed.mutex); feed.stop = true; } feed.changed.notify_all(); if (thread.joinable()) thread.join(); } } joiner = { reader, feed }; // (on every way out)
	double pushing = 0;
	for (unsigned int push = 0; ; ++push) {
		ahost_bam_piece piece;
		{ std::unique_lock<std::mutex> lock(feed.mutex); feed.changed.wait(lock, [&] { return feed.read > push || feed.at_end; }); if (feed.read <= push) break; piece = feed.pieces[push % buffers]; }
		const double before = now_seconds();
		if (piece.stored_bgzf) device_check(agpu_ingest_push_bgzf(run.device, run.pieces[push % buffers], piece.bytes, run.tables[push % buffers], piece.n_blocks, piece.stream_bytes));
		else device_check(agpu_ingest_push(run.device, run.pieces[push % buffers], piece.bytes));
		pushing += now_seconds() - before;
		{ std::lock_guard<std::mutex> lock(feed.mutex); feed.pushed = push + 1; }
		feed.changed.notify_all();
	}
	reader.join();
	if (!feed.error.empty()) throw Failure{ feed.error };
	run.feed_reading = feed.reading; run.feed_pushing = pushing; run.feed_finished = now_seconds();
	closer.armed = false; // (finish_device_ingest closes the file)
}

// One sample over several ranks: every rank has ingested its part of the records; ONE all-gather puts the batch of the whole sample on every rank (include/arriba_gpu.h:
// agpu_shard_export / agpu_shard_merge: fragments concatenate in name order, counters and coverage_t add up before saturation).  In device memory when the ranks hold an RCCL
// communicator, through the callbacks in host memory otherwise.  `result` becomes that of the whole sample.
void exchange_parts(Run& run, agpu_ingest_result& result) {
	const arriba_workflow_communicator& ranks = *run.ranks;
	const double started = now_seconds();
	if (ranks.rccl_communicator != nullptr) { // (size all-reduce, export, all-gather and merge in one call: a failure inside it is told behind it)
		together(run.ranks, [&] { device_check(agpu_shard_merge_rccl(run.device, ranks.rccl_communicator, ranks.size, &result)); });
		run.exchange_parts += now_seconds() - started;
		return;
	}
	uint64_t bytes = 0;
	together(run.ranks, [&] { device_check(agpu_shard_export_size(run.device, &bytes)); });
	int64_t widest = (int64_t) bytes; // the blocks travel at the stride of the largest part
	exchange_check(ranks.all_reduce_int64(ranks.state, &widest, 1, ARRIBA_WORKFLOW_MAX), "size of the parts");
	const uint64_t stride = ((uint64_t) widest + 15) & ~(uint64_t) 15;
	std::unique_ptr<uint8_t[]> mine, all;
	together(run.ranks, [&] { mine.reset(new uint8_t[stride]); all.reset(new uint8_t[(size_t) ranks.size * stride]); memset(mine.get(), 0, stride); device_check(agpu_shard_export(run.device, mine.get(), stride)); });
	exchange_check(ranks.all_gather(ranks.state, mine.get(), all.get(), stride), "parts of the sample");
	together(run.ranks, [&] { device_check(agpu_shard_merge(run.device, all.get(), stride, ranks.size, &result)); });
	run.exchange_parts += now_seconds() - started;
}

// ---- One sample over several ranks, the READS SHARDED (include/arriba_gpu.h, "One sample over the GPUs of a node, the READS SHARDED"; BASELINE.json north_star: the packed batch
// scattered across the GPUs, ONE all-gather of the emissions of find_fusions).  Rank r keeps the fragments of its part of the file; what the reference's result depends on across reads
// travels -- sums of counters and of coverage_t, the dummy genes, the winners of the duplicate keys, the first mate-gap samples and strandedness votes in name order, the emissions,
// one byte of state per fragment whenever the filters have changed, the verdict inputs of filter_in_vitro, the rows of the reads of the written candidates -- and the batch does not.

// the bytes of every rank, in rank order (the collectives of a communicator move equal shares: padded to the largest); sizes: what every rank gave
void gather_bytes(Run& run, const void* mine, uint64_t bytes, std::vector<uint8_t>& all, std::vector<uint64_t>& sizes, const char* what) {
	const arriba_workflow_communicator& ranks = *run.ranks;
	const double started = now_seconds();
	std::vector<int64_t> given(ranks.size, 0);
	given[ranks.rank] = (int64_t) bytes;
	exchange_check(ranks.all_reduce_int64(ranks.state, given.data(), ranks.size, ARRIBA_WORKFLOW_SUM), what);
	uint64_t width = 16, total = 0;
	sizes.assign(ranks.size, 0);
	for (uint32_t r = 0; r < ranks.size; ++r) { sizes[r] = (uint64_t) given[r]; width = std::max<uint64_t>(width, (sizes[r] + 15) & ~(uint64_t) 15); total += sizes[r]; }
	std::unique_ptr<uint8_t[]> padded, received;
	together(run.ranks, [&] { padded.reset(new uint8_t[width]); received.reset(new uint8_t[(size_t) ranks.size * width]); if (bytes > 0) memcpy(padded.get(), mine, bytes); if (bytes < width) memset(padded.get() + bytes, 0, width - bytes); all.resize(total); });
	exchange_check(ranks.all_gather(ranks.state, padded.get(), received.get(), width), what);
	uint64_t at = 0;
	for (uint32_t r = 0; r < ranks.size; ++r) { if (sizes[r] > 0) memcpy(all.data() + at, received.get() + (size_t) r * width, sizes[r]); at += sizes[r]; }
	run.exchanged_bytes += total - bytes;
	run.exchange_parts += now_seconds() - started;
}
void sum_over_ranks(Run& run, int64_t* values, uint64_t count, const char* what) {
	const double started = now_seconds();
	exchange_check(run.ranks->all_reduce_int64(run.ranks->state, values, count, ARRIBA_WORKFLOW_SUM), what);
	run.exchange_parts += now_seconds() - started;
}
uint64_t sum_over_ranks(Run& run, uint64_t value, const char* what) { int64_t word = (int64_t) value; sum_over_ranks(run, &word, 1, what); return (uint64_t) word; }

// ... of what a kernel wrote and a kernel will read (the emissions of find_fusions: 36 bytes per read and gene pair, 4.7 GB at 10^8 fragments; the states of the reads; the winners of
// the duplicate keys).  `fill` writes this rank's bytes to the pointer it is given; the result is the bytes of all ranks in rank order.  The ranks hold an RCCL communicator: the
// pointers are device memory (agpu_scratch_buffer) and ncclAllGather moves them over xGMI, nothing crosses PCIe; otherwise host memory through the callbacks.
struct Gathered { const uint8_t* data; uint64_t bytes; };
template <class Fill> Gathered gather_from_device(Run& run, uint64_t bytes, Fill fill, const char* what) {
	const arriba_workflow_communicator& ranks = *run.ranks;
	if (ranks.rccl_communicator == nullptr) {
		std::vector<uint8_t>& mine = run.exchange_mine; std::vector<uint64_t> sizes;
		together(run.ranks, [&] { mine.resize(bytes > 0 ? bytes : 1); if (bytes > 0) fill((void*) mine.data()); });
		gather_bytes(run, mine.data(), bytes, run.exchange_all, sizes, what);
		Gathered all = { run.exchange_all.data(), run.exchange_all.size() };
		return all;
	}
	const double started = now_seconds();
	std::vector<int64_t> given(ranks.size, 0);
	given[ranks.rank] = (int64_t) bytes;
	exchange_check(ranks.all_reduce_int64(ranks.state, given.data(), ranks.size, ARRIBA_WORKFLOW_SUM), what);
	uint64_t width = 16, total = 0;
	for (uint32_t r = 0; r < ranks.size; ++r) { width = std::max<uint64_t>(width, ((uint64_t) given[r] + 15) & ~(uint64_t) 15); total += (uint64_t) given[r]; }
	void* mine = nullptr; void* received = nullptr; void* all = nullptr;
	together(run.ranks, [&] { // (local: the buffers -- the likeliest place to run out of memory -- and this rank's bytes)
		device_check(agpu_scratch_buffer(run.device, "exchange.mine", width, &mine)); device_check(agpu_scratch_buffer(run.device, "exchange.received", (uint64_t) ranks.size * width, &received));
		device_check(agpu_scratch_buffer(run.device, "exchange.all", total, &all));
		if (bytes > 0) fill(mine);
	});
	device_check(agpu_rccl_all_gather_device(run.device, ranks.rccl_communicator, mine, received, width));
	together(run.ranks, [&] { uint64_t at = 0; for (uint32_t r = 0; r < ranks.size; ++r) { device_check(agpu_device_copy(run.device, (uint8_t*) all + at, (const uint8_t*) received + (size_t) r * width, (uint64_t) given[r])); at += (uint64_t) given[r]; } });
	run.exchanged_bytes += total - bytes;
	run.exchange_parts += now_seconds() - started;
	Gathered result = { (const uint8_t*) all, total };
	return result;
}

// what a walk over read lists asks of a read, of every fragment of the sample on every rank: one byte per fragment (agpu_read_state_export / _import), exchanged whenever a stage
// has changed the filters of reads where they live
void exchange_read_state(Run& run) {
	const Gathered all = gather_from_device(run, run.local_fragments, [&](void* mine) { device_check(agpu_read_state_export(run.device, (uint8_t*) mine)); }, "states of the reads");
	together(run.ranks, [&] {
		if (all.bytes != run.n_fragments) throw Failure{ "ERROR: the states of the reads of the ranks do not add up to the fragments of the sample" };
		device_check(agpu_read_state_import(run.device, all.data));
	});
}

// Behind agpu_ingest_finish of this rank's part.  true: the parts follow each other in name order (every part is sorted: no read name is in two of them) -- this rank keeps its
// fragments, `result` becomes that of the sample, coverage_t and the viral read counts of the sample are on the device and with the host session.  false: the names of the file
// are in another order (STAR writes the reads in the order of the FASTQ file when it is not asked to sort) -- the caller puts the batch together on every rank (exchange_parts).
bool shard_reads(Run& run, agpu_ingest_result& result, uint64_t windows, uint32_t n_contigs) {
	const arriba_workflow_communicator& ranks = *run.ranks;
	const char* knob = getenv("ARRIBA_RANKS_SPLIT"); // "replicated": the split of rounds 2-5 (one all-gather of the batch, the stages on every rank), for comparisons
	const bool wanted = !(knob != nullptr && strcmp(knob, "replicated") == 0);
	enum { NAME_BYTES = 512 };
	struct Boundary { char first[NAME_BYTES], last[NAME_BYTES]; uint64_t fragments; uint64_t wanted; } mine;
	std::vector<uint8_t> all; std::vector<uint64_t> sizes;
	together(run.ranks, [&] { memset(&mine, 0, sizeof(mine)); mine.fragments = result.fragments; mine.wanted = wanted ? 1 : 0; device_check(agpu_shard_boundary_names(run.device, mine.first, mine.last, NAME_BYTES)); });
	gather_bytes(run, &mine, sizeof(mine), all, sizes, "names at the ends of the parts");
	const Boundary* parts = (const Boundary*) all.data();
	bool in_order = true; std::string previous; bool have_previous = false;
	uint64_t first_rank = 0, global_n = 0;
	for (uint32_t r = 0; r < ranks.size; ++r) {
		if (!parts[r].wanted) in_order = false;
		if (r < ranks.rank) first_rank += parts[r].fragments;
		global_n += parts[r].fragments;
		if (parts[r].fragments == 0) continue;
		if (have_previous && !(previous < std::string(parts[r].first))) in_order = false; // (the order of std::string: the reference's std::map)
		previous = parts[r].last; have_previous = true;
	}
	if (!in_order) return false;
	// the counters of the parts add up (source/read_chimeric_alignments.cpp:560-773 leaves sums behind)
	int64_t sums[7] = { (int64_t) result.records, (int64_t) result.mapped_reads, (int64_t) result.malformed_count, (int64_t) result.missing_hi_tag, (int64_t) result.stream_bytes, result.names_were_sorted ? 0 : 1, result.no_chimeric_reads ? 0 : 1 };
	sum_over_ranks(run, sums, 7, "counters of the parts");
	std::vector<uint32_t> coverage; std::vector<uint8_t> starts, ends; std::vector<uint64_t> viral; std::vector<int64_t> words;
	together(run.ranks, [&] {
		device_check(agpu_shard_keep(run.device, first_rank, global_n));
		coverage.assign(windows > 0 ? windows : 1, 0); starts.assign(coverage.size(), 0); ends.assign(coverage.size(), 0); viral.assign(n_contigs > 0 ? n_contigs : 1, 0);
		device_check(agpu_coverage_partial(run.device, coverage.data(), starts.data(), ends.data(), viral.data()));
		// coverage_t adds up before its 16-bit saturation: the windows of a part, saturated at 65535 (min(sum of min(x, 65535), 65535) == min(sum x, 65535)), two to a 64-bit word
		// of the all-reduce -- the sum of the low halves stays below 2^32 for up to 65 536 parts; the viral read counts behind them
		words.assign((windows + 1) / 2 + n_contigs, 0);
		for (uint64_t w = 0; w < windows; ++w) words[w / 2] |= (int64_t) ((uint64_t) coverage[w] << (32 * (w & 1)));
		for (uint32_t c = 0; c < n_contigs; ++c) words[(windows + 1) / 2 + c] = (int64_t) viral[c];
	});
	if (!words.empty()) sum_over_ranks(run, words.data(), words.size(), "coverage of the parts");
	{ const double started = now_seconds();
	  if (windows > 0) { exchange_check(ranks.all_reduce_max_bytes(ranks.state, starts.data(), windows), "fragment starts of the parts"); exchange_check(ranks.all_reduce_max_bytes(ranks.state, ends.data(), windows), "fragment ends of the parts"); }
	  run.exchange_parts += now_seconds() - started; }
	together(run.ranks, [&] {
		for (uint64_t w = 0; w < windows; ++w) coverage[w] = (uint32_t) ((uint64_t) words[w / 2] >> (32 * (w & 1)));
		for (uint32_t c = 0; c < n_contigs; ++c) viral[c] = (uint64_t) words[(windows + 1) / 2 + c];
		device_check(agpu_coverage_total(run.device, coverage.data(), starts.data(), ends.data(), viral.data()));
	});
	run.sharded = true; run.first_rank = first_rank; run.local_fragments = result.fragments;
	result.fragments = global_n; result.records = (uint64_t) sums[0]; result.mapped_reads = (uint64_t) sums[1]; result.malformed_count = (uint64_t) sums[2]; result.missing_hi_tag = (uint64_t) sums[3];
	result.stream_bytes = (uint64_t) sums[4]; result.names_were_sorted = sums[5] == 0; result.no_chimeric_reads = sums[6] == 0;
	return true;
}

// ... and what is left of read_chimeric_alignments behind the last piece (agpu_ingest_finish), the counters, coverage_t and viral read counts back to the host session
void finish_device_ingest(Run& run, double waited_since) {
	struct Closer { Run& run; ~Closer() { if (run.bam_open) { ahost_bam_close(run.host); run.bam_open = false; } } } closer = { run };
	const uint64_t windows = run.coverage_windows; const uint32_t n_contigs = run.bam_contigs;
	const double fed = now_seconds();
	agpu_ingest_result result;
	together(run.ranks, [&] { device_check(agpu_ingest_finish(run.device, &result)); });
	run.sharded = false; run.first_rank = 0; run.exchanged_bytes = 0;
	if (run.ranks != nullptr && !shard_reads(run, result, windows, n_contigs)) exchange_parts(run, result);
	if (!run.sharded) run.local_fragments = result.fragments;
	if (run.after_ingest) run.after_ingest();
	const double finished = now_seconds();
	together(run.ranks, [&] {
		std::vector<uint64_t> viral(n_contigs > 0 ? n_contigs : 1);
		std::vector<uint16_t> coverage(windows > 0 ? windows : 1);
		std::vector<uint8_t> starts(coverage.size()), ends(coverage.size());
		device_check(agpu_get_viral_read_counts(run.device, viral.data()));
		device_check(agpu_get_coverage(run.device, coverage.data(), starts.data(), ends.data()));
		host_check(ahost_adopt_device_ingest(run.host, &result, viral.data(), coverage.data(), starts.data(), ends.data()));
	});
	run.n_fragments = result.fragments;
	run.ingest_records = result.records; run.ingest_stream_bytes = result.stream_bytes; run.ingest_fed = fed; run.ingest_finished = finished; run.ingest_adopted = now_seconds();
	(void) waited_since;
}
// ... and what it found, for the report and the timing of the call that asked for the sample (waited_since: when that call began to wait)
void note_device_ingest(Run& run, double waited_since) {
	if (run.timing) {
		run.timing->feed_read = run.feed_reading; run.timing->feed_push = run.feed_pushing; run.timing->feed_total = run.feed_finished - run.feed_started;
		if (run.ingest_finished_ahead) { run.timing->feed = now_seconds() - waited_since; run.timing->ingest = 0; run.timing->adopt = 0; } // (all of it beside the sample in front: what is left is the wait for it)
		else { run.timing->feed = run.ingest_fed - waited_since; run.timing->ingest = run.ingest_finished - run.ingest_fed; run.timing->adopt = run.ingest_adopted - run.ingest_finished; }
	}
	run.note("bam_records", run.ingest_records); // (for the report only: no line of the reference's log)
	run.note("bam_stream_bytes", run.ingest_stream_bytes);
}

// the writer prints names, CIGAR-derived pileups and sequences of the supporting reads of the candidates it writes: their rows come back from the device, and their filters.
// in_list_order (fusions.tsv): one row per entry of the read lists, in their order -- the reads of a candidate lie next to each other in every column, which is how the writer
// walks them; otherwise (discarded.tsv with -X: millions of candidates that share their discordant mates) one row per fragment, in fragment order.
void fetch_rows_for_writer(Run& run, ahost_fusion_table& table, int write_discarded, bool in_list_order) {
	if (!run.device_ingest) return;
	const bool profile = getenv("ARRIBA_WRITER_PROFILE") != nullptr;
	double mark = now_seconds();
	auto lap = [&](const char* what) { if (profile) { const double now = now_seconds(); fprintf(stderr, "[rows] %s: %.3f s\n", what, now - mark); mark = now; } };
	uint64_t count = 0;
	const uint32_t* fragments = nullptr;
	if (in_list_order) { count = table.list_offset[3 * (size_t) table.n_candidates]; fragments = table.read_lists; }
	else {
		host_check(ahost_fusion_table_reads(&table, write_discarded, nullptr, 0, &count));
		uint32_t* unique = run.stage<uint32_t>("rows.fragments", count);
		host_check(ahost_fusion_table_reads(&table, write_discarded, unique, count, &count));
		fragments = unique;
	}
	lap("fragments of the rows");
	uint64_t cigar_words = 0, sequence_bytes = 0, name_bytes = 0;
	device_check(agpu_gather_rows_begin(run.device, fragments, count, &cigar_words, &sequence_bytes, &name_bytes));
	lap("agpu_gather_rows_begin");
	agpu_batch_rows rows;
	memset(&rows, 0, sizeof(rows));
	rows.n_aln = run.stage<uint8_t>("rows.n_aln", count); rows.fbits = run.stage<uint8_t>("rows.fbits", count); rows.group = run.stage<uint32_t>("rows.group", count);
	static const char* const names[3][6] = { { "rows.contig0", "rows.start0", "rows.end0", "rows.abits0", "rows.cigar_offset0", "rows.cigar_count0" }, { "rows.contig1", "rows.start1", "rows.end1", "rows.abits1", "rows.cigar_offset1", "rows.cigar_count1" },
	                                         { "rows.contig2", "rows.start2", "rows.end2", "rows.abits2", "rows.cigar_offset2", "rows.cigar_count2" } };
	for (int k = 0; k < 3; ++k) {
		rows.contig[k] = run.stage<uint16_t>(names[k][0], count); rows.start[k] = run.stage<int32_t>(names[k][1], count); rows.end[k] = run.stage<int32_t>(names[k][2], count);
		rows.abits[k] = run.stage<uint8_t>(names[k][3], count); rows.cigar_offset[k] = run.stage<uint32_t>(names[k][4], count); rows.cigar_count[k] = run.stage<uint16_t>(names[k][5], count);
	}
	rows.seq_offset[0] = run.stage<uint32_t>("rows.seq_offset0", count); rows.seq_length[0] = run.stage<uint32_t>("rows.seq_length0", count);
	rows.seq_offset[1] = run.stage<uint32_t>("rows.seq_offset1", count); rows.seq_length[1] = run.stage<uint32_t>("rows.seq_length1", count);
	rows.cigar_pool = run.stage<uint32_t>("rows.cigar_pool", cigar_words + 1); rows.seq_pool = run.stage<uint8_t>("rows.seq_pool", sequence_bytes + 4);
	rows.name_offset = run.stage<uint32_t>("rows.name_offset", count + 1); rows.names = run.stage<char>("rows.names", name_bytes + 1);
	lap("staging buffers");
	device_check(agpu_gather_rows_copy(run.device, &rows));
	lap("agpu_gather_rows_copy");
	uint8_t* filters = run.stage<uint8_t>("rows.filter", count); // the filters of these fragments only, instead of one byte for every fragment of the sample
	device_check(agpu_get_filters_of(run.device, fragments, count, filters));
	table.read_filter_of_rows = filters;
	lap("agpu_get_filters_of");
	host_check(ahost_set_batch_rows(run.host, &rows, in_list_order ? nullptr : (count > 0 ? fragments : nullptr)));
	lap("ahost_set_batch_rows");
	if (profile) fprintf(stderr, "[rows] %llu rows, %llu CIGAR words, %llu sequence bytes, %llu name bytes\n", (unsigned long long) count, (unsigned long long) cigar_words, (unsigned long long) sequence_bytes, (unsigned long long) name_bytes);
}

// ... the reads of the sample sharded over the ranks: a row lives where its read does.  Every rank gathers the rows of the entries whose fragment it holds (agpu_gather_rows_* with
// its own fragment numbers), the rows travel with the number of their entry, and every rank puts them in the order of the entries -- the batch fetch_rows_for_writer hands to the
// host library on one rank.  (Every rank takes all rows although it formats every size-th row of the file: the lists of the written candidates are a few hundred thousand reads.)
void fetch_rows_from_their_ranks(Run& run, ahost_fusion_table& table, int write_discarded, bool in_list_order) {
	uint64_t count = 0; const uint32_t* fragments = nullptr;
	std::vector<uint32_t> local; std::vector<uint64_t> entry_of; std::vector<uint8_t> block, all; std::vector<uint64_t> sizes;
	struct Header { uint64_t rows, cigar_words, sequence_bytes, name_bytes; };
	// the layout of a block: Header, entry[rows] (u64), n_aln, fbits (u8), group (u32), per slot contig (u16) start end (i32) abits (u8) cigar_offset (u32) cigar_count (u16), seq_offset / seq_length x 2 (u32),
	// name_offset[rows + 1] (u32), cigar pool (u32), sequence pool (u8), names; every array on an 8-byte boundary
	auto aligned = [](uint64_t bytes) { return (bytes + 7) & ~(uint64_t) 7; };
	together(run.ranks, [&] {
		if (in_list_order) { count = table.list_offset[3 * (size_t) table.n_candidates]; fragments = table.read_lists; }
		else {
			host_check(ahost_fusion_table_reads(&table, write_discarded, nullptr, 0, &count));
			uint32_t* unique = run.stage<uint32_t>("rows.fragments", count);
			host_check(ahost_fusion_table_reads(&table, write_discarded, unique, count, &count));
			fragments = unique;
		}
		for (uint64_t k = 0; k < count; ++k) { const uint64_t at = (uint64_t) fragments[k] - run.first_rank; if (at < run.local_fragments) { local.push_back((uint32_t) at); entry_of.push_back(k); } }
		const uint64_t m = local.size();
		Header header = { m, 0, 0, 0 };
		static const uint32_t none = 0; // (a null pointer means "all fragments" to that call)
		device_check(agpu_gather_rows_begin(run.device, m > 0 ? local.data() : &none, m, &header.cigar_words, &header.sequence_bytes, &header.name_bytes));
		uint64_t bytes = sizeof(Header) + m * 8 + aligned(m) * 2 + aligned(m * 4) + 3 * (aligned(m * 2) * 2 + aligned(m * 4) * 3 + aligned(m)) + 4 * aligned(m * 4) + aligned((m + 1) * 4) + aligned((header.cigar_words + 1) * 4) + aligned(header.sequence_bytes + 4) + aligned(header.name_bytes + 1);
		block.assign(bytes, 0);
		uint8_t* at = block.data();
		auto take = [&](uint64_t size) { uint8_t* here = at; at += aligned(size); return here; };
		memcpy(take(sizeof(Header)), &header, sizeof(Header));
		if (m > 0) memcpy(take(m * 8), entry_of.data(), m * 8); else take(0);
		agpu_batch_rows rows; memset(&rows, 0, sizeof(rows));
		rows.n_aln = take(m); rows.fbits = take(m); rows.group = (uint32_t*) take(m * 4);
		for (int k = 0; k < 3; ++k) { rows.contig[k] = (uint16_t*) take(m * 2); rows.start[k] = (int32_t*) take(m * 4); rows.end[k] = (int32_t*) take(m * 4); rows.abits[k] = take(m); rows.cigar_offset[k] = (uint32_t*) take(m * 4); rows.cigar_count[k] = (uint16_t*) take(m * 2); }
		for (int k = 0; k < 2; ++k) { rows.seq_offset[k] = (uint32_t*) take(m * 4); rows.seq_length[k] = (uint32_t*) take(m * 4); }
		rows.name_offset = (uint32_t*) take((m + 1) * 4); rows.cigar_pool = (uint32_t*) take((header.cigar_words + 1) * 4); rows.seq_pool = take(header.sequence_bytes + 4); rows.names = (char*) take(header.name_bytes + 1);
		if (m > 0) device_check(agpu_gather_rows_copy(run.device, &rows));
	});
	const double before = now_seconds();
	gather_bytes(run, block.data(), block.size(), all, sizes, "rows of the supporting reads");
	run.exchange_rows += now_seconds() - before;
	together(run.ranks, [&] {
		// sizes of the pools of the whole: the rows of a rank keep their place inside the rank's pools, the pools of the ranks follow each other; the names are laid out in entry order
		uint64_t cigar_words = 0, sequence_bytes = 0, name_bytes = 0, rows_in_all = 0, at_block = 0;
		for (uint32_t r = 0; r < run.ranks->size; ++r) { if (sizes[r] < sizeof(Header)) throw Failure{ "ERROR: a block of rows of another rank is damaged" }; Header h; memcpy(&h, all.data() + at_block, sizeof(h)); cigar_words += h.cigar_words; sequence_bytes += (h.sequence_bytes + 3) & ~(uint64_t) 3; name_bytes += h.name_bytes; rows_in_all += h.rows; at_block += sizes[r]; }
		if (rows_in_all != count) throw Failure{ "ERROR: the rows of the ranks do not add up to the reads of the candidates that are written" };
		agpu_batch_rows rows; memset(&rows, 0, sizeof(rows));
		rows.n = count; rows.cigar_pool_size = cigar_words; rows.seq_pool_size = sequence_bytes; rows.names_size = name_bytes;
		rows.n_aln = run.stage<uint8_t>("rows.n_aln", count); rows.fbits = run.stage<uint8_t>("rows.fbits", count); rows.group = run.stage<uint32_t>("rows.group", count);
		static const char* const names[3][6] = { { "rows.contig0", "rows.start0", "rows.end0", "rows.abits0", "rows.cigar_offset0", "rows.cigar_count0" }, { "rows.contig1", "rows.start1", "rows.end1", "rows.abits1", "rows.cigar_offset1", "rows.cigar_count1" },
		                                         { "rows.contig2", "rows.start2", "rows.end2", "rows.abits2", "rows.cigar_offset2", "rows.cigar_count2" } };
		for (int k = 0; k < 3; ++k) {
			rows.contig[k] = run.stage<uint16_t>(names[k][0], count); rows.start[k] = run.stage<int32_t>(names[k][1], count); rows.end[k] = run.stage<int32_t>(names[k][2], count);
			rows.abits[k] = run.stage<uint8_t>(names[k][3], count); rows.cigar_offset[k] = run.stage<uint32_t>(names[k][4], count); rows.cigar_count[k] = run.stage<uint16_t>(names[k][5], count);
		}
		rows.seq_offset[0] = run.stage<uint32_t>("rows.seq_offset0", count); rows.seq_length[0] = run.stage<uint32_t>("rows.seq_length0", count);
		rows.seq_offset[1] = run.stage<uint32_t>("rows.seq_offset1", count); rows.seq_length[1] = run.stage<uint32_t>("rows.seq_length1", count);
		rows.cigar_pool = run.stage<uint32_t>("rows.cigar_pool", cigar_words + 1); rows.seq_pool = run.stage<uint8_t>("rows.seq_pool", sequence_bytes + 4);
		rows.name_offset = run.stage<uint32_t>("rows.name_offset", count + 1); rows.names = run.stage<char>("rows.names", name_bytes + 1);
		std::vector<uint32_t> name_length(count > 0 ? count : 1, 0);
		std::vector<const char*> name_of(count > 0 ? count : 1, nullptr);
		uint64_t cigar_base = 0, sequence_base = 0; at_block = 0;
		for (uint32_t r = 0; r < run.ranks->size; ++r) {
			const uint8_t* at = all.data() + at_block; at_block += sizes[r];
			auto take = [&](uint64_t size) { const uint8_t* here = at; at += aligned(size); return here; };
			Header h; memcpy(&h, take(sizeof(Header)), sizeof(h));
			const uint64_t m = h.rows;
			const uint64_t* entry = (const uint64_t*) take(m * 8);
			const uint8_t* n_aln = take(m); const uint8_t* fbits = take(m); const uint32_t* group = (const uint32_t*) take(m * 4);
			const uint16_t* contig[3]; const int32_t* start[3]; const int32_t* end[3]; const uint8_t* abits[3]; const uint32_t* cigar_offset[3]; const uint16_t* cigar_count[3]; const uint32_t* seq_offset[2]; const uint32_t* seq_length[2];
			for (int k = 0; k < 3; ++k) { contig[k] = (const uint16_t*) take(m * 2); start[k] = (const int32_t*) take(m * 4); end[k] = (const int32_t*) take(m * 4); abits[k] = take(m); cigar_offset[k] = (const uint32_t*) take(m * 4); cigar_count[k] = (const uint16_t*) take(m * 2); }
			for (int k = 0; k < 2; ++k) { seq_offset[k] = (const uint32_t*) take(m * 4); seq_length[k] = (const uint32_t*) take(m * 4); }
			const uint32_t* name_offset = (const uint32_t*) take((m + 1) * 4); const uint32_t* cigar_pool = (const uint32_t*) take((h.cigar_words + 1) * 4); const uint8_t* seq_pool = take(h.sequence_bytes + 4); const char* row_names = (const char*) take(h.name_bytes + 1);
			if ((uint64_t) (at - (all.data() + at_block - sizes[r])) > sizes[r]) throw Failure{ "ERROR: a block of rows of another rank is damaged" };
			for (uint64_t j = 0; j < m; ++j) {
				const uint64_t k = entry[j];
				if (k >= count) throw Failure{ "ERROR: a block of rows of another rank is damaged" };
				rows.n_aln[k] = n_aln[j]; rows.fbits[k] = fbits[j]; rows.group[k] = group[j];
				for (int slot = 0; slot < 3; ++slot) { rows.contig[slot][k] = contig[slot][j]; rows.start[slot][k] = start[slot][j]; rows.end[slot][k] = end[slot][j]; rows.abits[slot][k] = abits[slot][j]; rows.cigar_offset[slot][k] = cigar_offset[slot][j] + (uint32_t) cigar_base; rows.cigar_count[slot][k] = cigar_count[slot][j]; }
				for (int slot = 0; slot < 2; ++slot) { rows.seq_offset[slot][k] = seq_offset[slot][j] + (uint32_t) (sequence_base / 4); rows.seq_length[slot][k] = seq_length[slot][j]; }
				name_length[k] = name_offset[j + 1] - name_offset[j]; name_of[k] = row_names + name_offset[j];
			}
			if (h.cigar_words > 0) memcpy(rows.cigar_pool + cigar_base, cigar_pool, h.cigar_words * 4);
			if (h.sequence_bytes > 0) memcpy(rows.seq_pool + sequence_base, seq_pool, h.sequence_bytes);
			cigar_base += h.cigar_words; sequence_base += (h.sequence_bytes + 3) & ~(uint64_t) 3;
		}
		uint64_t name_at = 0;
		for (uint64_t k = 0; k < count; ++k) { rows.name_offset[k] = (uint32_t) name_at; if (name_length[k] > 0) memcpy(rows.names + name_at, name_of[k], name_length[k]); name_at += name_length[k]; }
		rows.name_offset[count] = (uint32_t) name_at;
		uint8_t* filters = run.stage<uint8_t>("rows.filter", count); // (from the replicated states of the reads: agpu_get_filters_of takes global name ranks here)
		device_check(agpu_get_filters_of(run.device, fragments, count, filters));
		table.read_filter_of_rows = filters;
		host_check(ahost_set_batch_rows(run.host, &rows, in_list_order ? nullptr : (count > 0 ? fragments : nullptr)));
	});
}

// One sample over several ranks: the rows of a file are independent of each other (the fusion transcripts from the pileups of the supporting reads are the expensive part), so
// rank r formats the rows r, r + size, r + 2 size, ... (ahost_format_fusions; the header line in front of the rows of rank 0), the texts are gathered and rank 0 writes row k of
// the file from the text of rank k % size.  (An all-gather: the collectives a caller must supply are few; the texts of fusions.tsv are a few megabytes.)
void write_file_over_ranks(Run& run, const ahost_fusion_table& table, const char* path, int write_discarded, int print_extra_info, int32_t max_mate_gap) {
	const arriba_workflow_communicator& ranks = *run.ranks;
	const char* text = nullptr; uint64_t bytes = 0;
	together(run.ranks, [&] { host_check(ahost_format_fusions(run.host, &table, write_discarded, print_extra_info, run.options.device.max_itd_length, max_mate_gap, run.options.fill_sequence_gaps, ranks.rank, ranks.size, &text, &bytes)); });
	const double before = now_seconds();
	std::vector<int64_t> sizes(ranks.size, 0);
	sizes[ranks.rank] = (int64_t) bytes;
	exchange_check(ranks.all_reduce_int64(ranks.state, sizes.data(), ranks.size, ARRIBA_WORKFLOW_SUM), "sizes of the row texts");
	uint64_t width = 1;
	for (uint32_t r = 0; r < ranks.size; ++r) width = std::max<uint64_t>(width, (uint64_t) sizes[r]);
	std::vector<char> mine, all;
	together(run.ranks, [&] { mine.assign(width, 0); all.resize((size_t) ranks.size * width); if (bytes > 0) memcpy(mine.data(), text, bytes); });
	exchange_check(ranks.all_gather(ranks.state, mine.data(), all.data(), width), "texts of the rows");
	run.exchange_rows += now_seconds() - before;
	together(run.ranks, [&] {
		if (ranks.rank != 0) return;
		FILE* out = fopen(path, "wb");
		if (!out) throw Failure{ std::string("ERROR: failed to open output file '") + path + "'" };
		struct Closer { FILE* file; ~Closer() { if (file) fclose(file); } } closer = { out };
		std::vector<const char*> at(ranks.size), end(ranks.size);
		for (uint32_t r = 0; r < ranks.size; ++r) { at[r] = all.data() + (size_t) r * width; end[r] = at[r] + sizes[r]; }
		bool ok = true;
		auto write_line = [&](uint32_t r) { // the next line of the text of rank r (every line ends with a newline); false: there is none
			if (at[r] >= end[r]) return false;
			const char* stop = (const char*) memchr(at[r], '\n', (size_t) (end[r] - at[r]));
			const size_t length = stop ? (size_t) (stop - at[r]) + 1 : (size_t) (end[r] - at[r]);
			ok = ok && fwrite(at[r], 1, length, out) == length;
			at[r] += length;
			return true;
		};
		write_line(0); // the header
		for (uint64_t row = 0; write_line((uint32_t) (row % ranks.size)); ++row) {}
		for (uint32_t r = 0; r < ranks.size; ++r) if (at[r] < end[r]) throw Failure{ "ERROR: the rows of the ranks do not interleave (a rank formatted more rows than its turn)" };
		closer.file = nullptr;
		if (fclose(out) != 0 || !ok) throw Failure{ std::string("ERROR: failed to write output file '") + path + "'" };
	});
}

// the output files: the device's results brought back, formatted by the host library (source/arriba.cpp:586-610).  Only the candidates a file will hold
// travel: fusions.tsv holds the few thousand that passed every filter, of millions -- they are picked on the device (agpu_select_candidates), with their read lists, the
// rows of their supporting reads and the filters of those.  discarded.tsv (-O) counts the discarded reads of every discarded candidate by filter: the same call, more rows.
void write_output_files(Run& run, int32_t max_mate_gap) {
	double mark = now_seconds();
	auto lap = [&](double arriba_workflow_timing::* part) { const double now = now_seconds(); if (run.timing) run.timing->*part += now - mark; mark = now; };
	const uint32_t n_genes = ahost_annotation_view(run.host)->n_genes + run.dummy_genes;
	std::vector<uint16_t>& gene_contig = run.writer_gene_contig; std::vector<int32_t>& gene_start = run.writer_gene_start; std::vector<int32_t>& gene_end = run.writer_gene_end;
	together(run.ranks, [&] {
	device_check(agpu_assign_confidence(run.device, nullptr)); // behind the 'isoforms' filter: recovered isoforms are scored anew
	device_check(agpu_candidate_iteration_order(run.device, nullptr));
	gene_contig.assign(n_genes > 0 ? n_genes : 1, 0); gene_start.assign(gene_contig.size(), 0); gene_end.assign(gene_contig.size(), 0);
	device_check(agpu_get_gene_table(run.device, 0, n_genes, gene_contig.data(), gene_start.data(), gene_end.data(), nullptr, nullptr));
	if (run.options.tags_file && !run.tags_loaded) { run.say(std::string("Loading tags from '") + run.options.tags_file + "'"); host_check(ahost_load_tags(run.host, run.options.tags_file)); run.tags_loaded = true; } // (once per session)
	if (run.options.protein_domains_file && !run.domains_loaded) { run.say(std::string("Loading protein domains from '") + run.options.protein_domains_file + "'"); host_check(ahost_load_protein_domains(run.host, run.options.protein_domains_file)); run.domains_loaded = true; }
	});

	for (int write_discarded = 0; write_discarded <= (run.options.discarded_output_file ? 1 : 0); ++write_discarded) {
		ahost_fusion_table table;
		const int print_extra_info = write_discarded ? run.options.print_extra_info_for_discarded_fusions : 1;
		together(run.ranks, [&] { // (several ranks: every rank holds the same candidates and fetches the same rows; what differs is which rows it formats)
		uint64_t w = 0;
		device_check(agpu_select_candidates(run.device, write_discarded, &w));
		agpu_selected_candidates columns;
		columns.candidate = run.stage<uint32_t>("table.candidate", w); columns.gene1 = run.stage<uint32_t>("table.gene1", w); columns.gene2 = run.stage<uint32_t>("table.gene2", w); columns.contigs = run.stage<uint32_t>("table.contigs", w);
		columns.breakpoint1 = run.stage<int32_t>("table.breakpoint1", w); columns.breakpoint2 = run.stage<int32_t>("table.breakpoint2", w); columns.flags = run.stage<uint32_t>("table.flags", w); columns.filter = run.stage<uint8_t>("table.filter", w);
		columns.split_reads1 = run.stage<uint32_t>("table.split_reads1", w); columns.split_reads2 = run.stage<uint32_t>("table.split_reads2", w); columns.discordant_mates = run.stage<uint32_t>("table.discordant_mates", w);
		columns.evalue = run.stage<float>("table.evalue", w); columns.confidence = run.stage<uint8_t>("table.confidence", w); columns.iteration_rank = run.stage<uint32_t>("table.iteration_rank", w);
		columns.closest_genomic_breakpoint1 = run.stage<int32_t>("table.closest1", w); columns.closest_genomic_breakpoint2 = run.stage<int32_t>("table.closest2", w);
		device_check(agpu_get_selected_candidates(run.device, &columns));
		uint64_t* list_offset = run.stage<uint64_t>("table.list_offset", 3 * w + 1);
		list_offset[0] = 0;
		uint64_t total = 0;
		device_check(agpu_get_candidate_read_lists_of(run.device, columns.candidate, w, list_offset, nullptr, 0, &total));
		uint32_t* read_lists = run.stage<uint32_t>("table.read_lists", total);
		if (total > 0) device_check(agpu_get_candidate_read_lists_of(run.device, columns.candidate, w, list_offset, read_lists, total, &total));
		memset(&table, 0, sizeof(table));
		table.n_candidates = (uint32_t) w;
		table.gene1 = columns.gene1; table.gene2 = columns.gene2; table.contigs = columns.contigs; table.breakpoint1 = columns.breakpoint1; table.breakpoint2 = columns.breakpoint2; table.flags = columns.flags; table.filter = columns.filter;
		table.split_reads1 = columns.split_reads1; table.split_reads2 = columns.split_reads2; table.discordant_mates = columns.discordant_mates; table.list_offset = list_offset; table.read_lists = read_lists;
		table.evalue = columns.evalue; table.confidence = columns.confidence; table.iteration_rank = columns.iteration_rank;
		table.closest_genomic_breakpoint1 = columns.closest_genomic_breakpoint1; table.closest_genomic_breakpoint2 = columns.closest_genomic_breakpoint2;
		table.n_genes = n_genes; table.gene_contig = gene_contig.data(); table.gene_start = gene_start.data(); table.gene_end = gene_end.data();
		if (write_discarded) run.say(std::string("Writing discarded fusions to file '") + run.options.discarded_output_file + "'");
		else run.say(std::string("Writing fusions to file '") + run.options.output_file + "' ");
		const bool rows_from_device = print_extra_info && run.device_ingest;
		if (!rows_from_device) { // the filter of every fragment: the writer counts the discarded reads of a candidate by filter (and, after a host ingest, works on the batch of the session)
			uint8_t* read_filter = run.stage<uint8_t>("table.read_filter", run.n_fragments);
			device_check(agpu_get_filters(run.device, read_filter));
			table.read_filter = read_filter;
		}
		lap(&arriba_workflow_timing::output_results);
		if (rows_from_device && !run.sharded) fetch_rows_for_writer(run, table, write_discarded, write_discarded == 0);
		});
		if (print_extra_info && run.device_ingest && run.sharded) fetch_rows_from_their_ranks(run, table, write_discarded, write_discarded == 0); // (an exchange: behind the status of the block above)
		lap(&arriba_workflow_timing::output_rows);
		together(run.ranks, [&] { if (run.before_host_writer) run.before_host_writer(); });
		if (run.ranks != nullptr) { // rank r formats the rows r, r + size, ... of the file; rank 0 gathers the texts, puts the rows back in order and writes
			write_file_over_ranks(run, table, write_discarded ? run.options.discarded_output_file : run.options.output_file, write_discarded, print_extra_info, max_mate_gap);
			lap(&arriba_workflow_timing::output_format);
			continue;
		}
		const bool last_file = write_discarded == (run.options.discarded_output_file ? 1 : 0);
		if (run.defer_output && last_file) { // nothing of this file is on the device any more: formatted and written beside the next sample
			const std::string path = write_discarded ? run.options.discarded_output_file : run.options.output_file;
			const unsigned int max_itd_length = run.options.device.max_itd_length; const int fill_gaps = run.options.fill_sequence_gaps;
			run.writer_error.clear();
			Run* lane = &run;
			// what the writer reads of the sample leaves the host session (ahost_detach_sample): the feed of the lane's next sample may open its file at once.  (Until round 4b it
			// waited for this file -- 0.43 s of writer against the 0.2 s after which the stream buffer is free: every other step of a queue of 10^8-fragment samples waited 0.2-0.4 s
			// for its feed, profiles/r04g2_bench100m_steps.txt.)  The tables of `table` are the lane's staging buffers: nothing touches them before the lane's next sample is worked
			// on, and arriba_workflow_sample joins this thread first.
			ahost_detached_sample* sample = ahost_detach_sample(run.host);
			if (!sample) throw Failure{ std::string("ERROR: ") + ahost_last_error() };
			run.writer = std::thread([lane, sample, table, path, write_discarded, print_extra_info, max_itd_length, max_mate_gap, fill_gaps] {
				const unsigned int budget = ahost_cpu_budget();
				// (beside the feed of the next sample and the thread that runs its stages.  As long as the next feed on this lane waited for this file the writer was not to be the slow one: with 6 of
				// 16 threads it took 0.56 s and that feed started 0.18 s late, a 10^8-fragment step 2.22 s; with 10: 0.37 s, 2.08 s, profiles/r04r_*.  Since the writer works from a detached
				// sample nobody waits for it before the lane's next sample is worked on, two steps later, and it is the one to stand back again: 8 readers of the feed + 6 formatters + the two
				// threads that talk to the device are the 16 CPUs of the quota, and more busy threads than that are all stopped together.  Its threads format at nice 10: worker_thread_starts.)
				ahost_limit_threads_of_this_thread(std::max(2u, budget * 3 / 8));
				const double started = now_seconds();
				if (ahost_write_fusions_of(sample, &table, path.c_str(), write_discarded, print_extra_info, max_itd_length, max_mate_gap, fill_gaps) != 0) lane->writer_error = std::string("ERROR: ") + ahost_last_error();
				ahost_release_sample(sample);
				lane->writer_seconds = now_seconds() - started;
			});
			lap(&arriba_workflow_timing::output_format);
			continue;
		}
		host_check(ahost_write_fusions(run.host, &table, write_discarded ? run.options.discarded_output_file : run.options.output_file, write_discarded, print_extra_info, run.options.device.max_itd_length, max_mate_gap, run.options.fill_sequence_gaps));
		lap(&arriba_workflow_timing::output_format);
	}
}

// filter_mismappers (source/filter_mismappers.cpp:272-359) over the ranks that hold the same batch: the re-alignments are independent per read -- rank r takes the jobs
// r, r + size, ... of the list every rank builds alike --, ONE all-reduce (max) of the verdict bytes, then every rank discards the reads and judges the candidates
void filter_mismappers_over_ranks(Run& run, int32_t max_mate_gap, uint64_t* remaining, uint64_t* discarded_reads) {
	const arriba_workflow_communicator& ranks = *run.ranks;
	if (ranks.rccl_communicator != nullptr) { // (jobs, verdicts, all-reduce in device memory and apply in one call)
		together(run.ranks, [&] { device_check(agpu_filter_mismappers_rccl(run.device, ranks.rccl_communicator, max_mate_gap, ranks.rank, ranks.size, remaining, discarded_reads)); });
		return;
	}
	uint64_t n_jobs = 0;
	std::vector<uint8_t> verdicts;
	together(run.ranks, [&] {
		device_check(agpu_mismapper_jobs(run.device, &n_jobs));
		verdicts.assign(n_jobs > 0 ? n_jobs : 1, 0);
		device_check(agpu_mismapper_verdicts(run.device, max_mate_gap, ranks.rank, ranks.size, verdicts.data()));
	});
	const double before = now_seconds();
	if (n_jobs > 0) exchange_check(ranks.all_reduce_max_bytes(ranks.state, verdicts.data(), n_jobs), "verdicts of filter_mismappers");
	run.exchange_verdicts = now_seconds() - before;
	together(run.ranks, [&] { device_check(agpu_filter_mismappers_apply(run.device, verdicts.data(), remaining, discarded_reads)); });
}

// what main() does once per process: assembly, annotation, index (source/arriba.cpp:97-113); the device context with the annotation in HBM
void open_session(Run& run) {
	const arriba_workflow_options& o = run.options;
	if (!o.assembly_file || !o.gene_annotation_file) throw Failure{ "ERROR: assembly, gene annotation, alignments and output file are required" };
	run.say(std::string("Loading assembly from '") + o.assembly_file + "' ");
	run.say(std::string("Loading annotation from '") + o.gene_annotation_file + "' ");
	run.host = ahost_open(o.assembly_file, o.gene_annotation_file, o.interesting_contigs, o.viral_contigs, o.gtf_features);
	if (!run.host) throw Failure{ std::string("ERROR: ") + ahost_last_error() };
	run.params = o.device;
	if (run.params.strandedness > 2) run.params.strandedness = 0; // resolved per sample
	run.device = agpu_create(o.device_index, &run.params);
	if (!run.device) throw Failure{ std::string("ERROR: ") + agpu_last_error() };
	device_check(agpu_upload_annotation(run.device, ahost_annotation_view(run.host)));
}

// what main() does per sample: read_chimeric_alignments ... the output files (source/arriba.cpp:119-610)
// what a sample starts with, in front of its feed (on the caller's thread: the feed may run on one of its own)
void prepare_sample(Run& run) {
	const arriba_workflow_options& o = run.options;
	if (!o.chimeric_bam_file) throw Failure{ "ERROR: assembly, gene annotation, alignments and output file are required" };
	agpu_params& params = run.params;
	params = o.device;
	if (params.strandedness > 2) params.strandedness = 0; // resolved in run_sample
	device_check(agpu_set_params(run.device, &params));
	run.dummy_genes = 0; run.n_candidates = 0; run.n_fragments = 0;
	run.device_ingest = !o.host_ingest;
}

// The stages of a sample whose reads are sharded over the ranks (shard_reads): source/arriba.cpp:141-584 in the reference's order.  What runs on the reads a rank holds, and what is
// exchanged in front of the next step, is said line by line; the stages that look at candidates alone run on every rank over the same table.  In front of every exchange the ranks
// tell each other how they fared (together()).
void run_stages_sharded(Run& run, agpu_params& params, int32_t& max_mate_gap, double& mismappers_seconds) {
	const arriba_workflow_options& o = run.options;
	const arriba_workflow_communicator& ranks = *run.ranks;
	uint64_t count = 0;
	std::vector<uint8_t> all; std::vector<uint64_t> sizes;
	// :141-143 mark_multimappers: the parts are cut between read names, no group of alignments is split
	together(run.ranks, [&] { device_check(agpu_mark_multimappers(run.device, &count)); });
	run.note("mark_multimappers", sum_over_ranks(run, count, "multi-mapping alignments"));
	// :145-163 detect_strandedness: the first 100 informative fragments of the sample in name order -- every rank says how many of them it could give, then counts its share
	if (o.device.strandedness <= 2) params.strandedness = o.device.strandedness;
	else {
		const uint32_t sample_size = 100; const float threshold = 0.95f; // (source/read_stats.cpp:96,135)
		uint32_t informative = 0, matching = 0;
		together(run.ranks, [&] { device_check(agpu_strandedness_votes(run.device, sample_size, &informative, &matching)); });
		std::vector<int64_t> given(ranks.size, 0); given[ranks.rank] = informative;
		sum_over_ranks(run, given.data(), ranks.size, "strandedness votes");
		uint64_t before = 0; for (uint32_t r = 0; r < ranks.rank; ++r) before += (uint64_t) given[r];
		const uint32_t share = before >= sample_size ? 0 : std::min<uint32_t>(informative, sample_size - (uint32_t) before);
		together(run.ranks, [&] { if (share < informative) device_check(agpu_strandedness_votes(run.device, share, &informative, &matching)); if (share == 0) { informative = 0; matching = 0; } });
		int64_t votes[2] = { informative, matching };
		sum_over_ranks(run, votes, 2, "strandedness votes");
		int verdict = 0;
		if ((uint64_t) votes[0] >= sample_size) { if ((float) votes[1] < (1 - threshold) * (float) votes[0]) verdict = 2; else if ((float) votes[1] > threshold * (float) votes[0]) verdict = 1; }
		params.strandedness = (uint8_t) verdict;
		run.say(std::string("Detecting strandedness (") + (params.strandedness == 1 ? "yes" : params.strandedness == 2 ? "reverse" : "no") + ")");
	}
	if (params.strandedness != 0) run.say("Assigning strands to alignments ");
	run.say("Annotating alignments ");
	// :187-325 annotate_alignments: the dummy genes are cut from the unmapped positions of the whole sample
	std::vector<uint64_t> positions; uint64_t n_unmapped = 0;
	together(run.ranks, [&] {
		device_check(agpu_set_params(run.device, &params));
		device_check(agpu_annotate_begin(run.device, &n_unmapped));
		positions.resize(n_unmapped > 0 ? n_unmapped : 1);
		device_check(agpu_copy_unmapped_positions(run.device, positions.data()));
	});
	gather_bytes(run, positions.data(), n_unmapped * 8, all, sizes, "positions without a gene");
	// :327-350 duplicates (the first fragment of a key in the name order of the SAMPLE stays), uninteresting and viral contigs (the verdicts per contig: sequential host work on the
	// integration sites of all ranks, coverage_t and the viral read counts of the sample)
	std::vector<uint32_t> pairs; uint64_t n_pairs = 0;
	together(run.ranks, [&] {
		device_check(agpu_annotate_finish(run.device, all.empty() ? nullptr : (const uint64_t*) all.data(), all.size() / 8, &run.dummy_genes));
		device_check(agpu_get_viral_integration_sites(run.device, nullptr, 0, &n_pairs));
		pairs.resize(2 * (n_pairs > 0 ? n_pairs : 1));
		device_check(agpu_get_viral_integration_sites(run.device, pairs.data(), n_pairs, &n_pairs));
	});
	gather_bytes(run, pairs.data(), n_pairs * 8, all, sizes, "viral integration sites");
	const uint32_t n_genes = ahost_annotation_view(run.host)->n_genes + run.dummy_genes, n_contigs = ahost_contig_count(run.host);
	std::vector<uint8_t> top(n_contigs > 0 ? n_contigs : 1), low(top.size()); uint64_t n_entries = 0;
	together(run.ranks, [&] {
		std::vector<uint8_t> gene_bits(n_genes > 0 ? n_genes : 1);
		device_check(agpu_get_gene_table(run.device, 0, n_genes, nullptr, nullptr, nullptr, gene_bits.data(), nullptr));
		host_check(ahost_viral_verdicts(run.host, (const uint32_t*) all.data(), all.size() / 8, gene_bits.data(), n_genes, o.top_viral_contigs, o.viral_contig_min_covered_fraction, top.data(), low.data()) < 0 ? -1 : 0);
		device_check(agpu_duplicates_begin(run.device, &n_entries));
	});
	const Gathered winners = gather_from_device(run, n_entries * AGPU_DUPLICATE_ENTRY_BYTES, [&](void* mine) { device_check(agpu_copy_duplicate_entries(run.device, mine)); }, "winners of the duplicate keys");
	// :352-364 estimate_fragment_length: the mate gaps of the first 100001 qualifying fragments in name order and the read lengths of every fragment the reference's loop visits on
	// the way, summed sequentially in float (hazards H4 / H6): shard by shard, every rank continuing the sum of the rank in front of it
	const uint32_t max_samples = 100001;
	std::vector<int32_t> mate_gaps(max_samples); uint32_t n_samples = 0; uint64_t visited = 0;
	together(run.ranks, [&] {
		device_check(agpu_read_filters_stage1_global(run.device, winners.bytes == 0 ? nullptr : winners.data, winners.bytes / AGPU_DUPLICATE_ENTRY_BYTES, top.data(), low.data()));
		device_check(agpu_fragment_length_samples_limited(run.device, max_samples, mate_gaps.data(), &n_samples, &visited));
	});
	uint64_t visited_here = 0; uint32_t wanted = 0;
	{ std::vector<int64_t> given(ranks.size, 0); given[ranks.rank] = n_samples;
	  sum_over_ranks(run, given.data(), ranks.size, "mate-gap samples");
	  uint64_t before = 0; for (uint32_t r = 0; r < ranks.rank; ++r) before += (uint64_t) given[r];
	  together(run.ranks, [&] {
		if (before >= max_samples) { wanted = 0; visited_here = 0; } // the loop stopped in a shard in front of this one
		else if (n_samples >= max_samples - before) { // ... stops in this one, right behind the fragment that delivers sample number 100001
			wanted = (uint32_t) (max_samples - before);
			if (wanted < max_samples) device_check(agpu_fragment_length_samples_limited(run.device, wanted, mate_gaps.data(), &n_samples, &visited));
			visited_here = visited;
		} else { wanted = n_samples; visited_here = run.local_fragments; } // ... visits the whole shard
	  }); }
	gather_bytes(run, mate_gaps.data(), (uint64_t) wanted * 4, all, sizes, "mate-gap samples");
	float read_length_sum = 0; uint64_t visited_in_all = 0;
	{ std::vector<uint32_t> length1, length2;
	  together(run.ranks, [&] { length1.resize(visited_here > 0 ? visited_here : 1); length2.resize(length1.size()); device_check(agpu_get_read_lengths(run.device, 0, visited_here, length1.data(), length2.data())); });
	  for (uint32_t r = 0; r < ranks.size; ++r) { // (a chain: rank r adds its lengths to the sum of the ranks in front of it; the others pass a zero)
		int64_t word = 0;
		if (r == ranks.rank) { const float mine = ahost_read_length_sum_of(read_length_sum, length1.data(), length2.data(), visited_here); memcpy(&word, &mine, sizeof(mine)); }
		sum_over_ranks(run, &word, 1, "sum of the read lengths");
		memcpy(&read_length_sum, &word, sizeof(read_length_sum));
	  }
	  visited_in_all = sum_over_ranks(run, visited_here, "fragments visited"); }
	float mate_gap_mean = 0, mate_gap_stddev = 0, read_length_mean = 0;
	const uint32_t samples_in_all = (uint32_t) (all.size() / 4);
	// :366-409 the read-level filters, each on the reads a rank holds; "(remaining=N)" is the sum
	std::vector<uint64_t> remaining(AGPU_FILTER_COUNT);
	together(run.ranks, [&] {
		if (ahost_estimate_fragment_length_from_sums((const int32_t*) all.data(), samples_in_all, read_length_sum, visited_in_all, params.fragment_length, &mate_gap_mean, &mate_gap_stddev, &read_length_mean, &max_mate_gap) < 0)
			throw Failure{ std::string("ERROR: ") + ahost_last_error() };
		device_check(agpu_read_filters_stage2(run.device, remaining.data()));
	});
	sum_over_ranks(run, (int64_t*) remaining.data(), AGPU_FILTER_COUNT, "fragments remaining behind the read filters");
	{ std::ostringstream fragment_length_line;
	  fragment_length_line << "Estimating fragment length ";
	  if (samples_in_all >= 10000) fragment_length_line << "(mate gap mean=" << mate_gap_mean << ", mate gap stddev=" << mate_gap_stddev << ", read length mean=" << read_length_mean << ")";
	  static const struct { unsigned id; const char* name; } read_filters[] = { { 1, "filter_duplicates" }, { 30, "filter_uninteresting_contigs" }, { 31, "filter_viral_contigs" }, { 32, "filter_top_expressed_viral_contigs" },
		{ 33, "filter_low_coverage_viral_contigs" }, { 4, "filter_proximal_read_through" }, { 2, "filter_inconsistently_clipped_mates" }, { 3, "filter_homopolymer" }, { 6, "filter_small_insert_size" }, { 7, "filter_long_gap" },
		{ 5, "filter_same_gene" }, { 8, "filter_hairpin" }, { 10, "filter_mismatches" }, { 36, "filter_low_entropy" } };
	  for (size_t f = 0; f < sizeof(read_filters) / sizeof(read_filters[0]); ++f) {
		if (f == 5) run.say(fragment_length_line.str());
		run.note(read_filters[f].name, remaining[read_filters[f].id]);
	  } }
	// :411-413 find_fusions: the emissions of the reads of every rank (one record per read x gene1 x gene2, name order), ONE all-gather, the candidates and their read lists built on
	// every rank from the emissions of all -- the same table everywhere, its lists in global name ranks
	uint64_t n_emissions = 0;
	together(run.ranks, [&] { device_check(agpu_build_emissions(run.device, 1, &n_emissions)); });
	const Gathered emissions = gather_from_device(run, n_emissions * AGPU_EMISSION_BYTES, [&](void* mine) { device_check(agpu_copy_emissions(run.device, mine)); }, "emissions of find_fusions");
	together(run.ranks, [&] {
		device_check(agpu_find_fusions_from_emissions(run.device, emissions.bytes == 0 ? nullptr : emissions.data, emissions.bytes / AGPU_EMISSION_BYTES, max_mate_gap, &count));
		run.n_candidates = count;
		if (o.log_to_stdout) {
			std::vector<uint8_t> filter(count > 0 ? count : 1);
			device_check(agpu_get_candidates(run.device, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, filter.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
			uint64_t unfiltered = 0;
			for (uint64_t c = 0; c < count; ++c) unfiltered += filter[c] == 0;
			run.progress("find_fusions", unfiltered);
		}
		if (run.report && run.report->n_stages < sizeof(run.report->stages) / sizeof(run.report->stages[0])) { arriba_workflow_stage& entry = run.report->stages[run.report->n_stages++]; snprintf(entry.stage, sizeof(entry.stage), "find_fusions"); entry.count = count; }
	});
	all.clear(); all.shrink_to_fit();
	exchange_read_state(run); // what the walks over the read lists ask of a read: the filters of the cascade, multi-mapper, exonic -- of every fragment of the sample
	// :415-460
	uint64_t discarded_reads = 0;
	together(run.ranks, [&] {
		if (o.genomic_breakpoints_file) {
			const agpu_genomic_breakpoint* variants = nullptr; uint32_t n_variants = 0;
			host_check(ahost_load_genomic_breakpoints(run.host, o.genomic_breakpoints_file, &variants, &n_variants));
			device_check(agpu_mark_genomic_support(run.device, variants, n_variants, o.max_genomic_breakpoint_distance, &count)); run.note("mark_genomic_support", count);
		}
		device_check(agpu_merge_adjacent_fusions(run.device, 5, &count)); run.note("merge_adjacent_fusions", count);
		// filter_multimappers: the best candidate of every multi-mapping read from the lists (all here), the alignment scores and the choice where the reads are ...
		device_check(agpu_filter_multimappers_resolve(run.device, &discarded_reads));
	});
	exchange_read_state(run); // ... the reads that lost, to every rank ...
	uint64_t discarded[3];
	together(run.ranks, [&] {
		device_check(agpu_filter_multimappers_recount(run.device, &count)); run.note("filter_multimappers", count); // ... and the counters of the candidates from the replicated states
		device_check(agpu_candidate_iteration_order(run.device, nullptr));
		run.say("Estimating expected number of fusions by random chance (e-value) ");
		device_check(agpu_estimate_expected_fusions(run.device, run.mapped_reads, nullptr));
		device_check(agpu_filter_candidate_predicates(run.device, discarded));
		if (o.log_to_stdout) {
			const uint64_t before = count;
			run.progress("filter_non_coding_neighbors", before - discarded[0]); run.progress("filter_intragenic_both_exonic", before - discarded[0] - discarded[1]); run.progress("filter_min_support", before - discarded[0] - discarded[1] - discarded[2]);
		}
		device_check(agpu_filter_relative_support(run.device, &count)); run.note("filter_relative_support", count);
		// :463-478 (recover_internal_tandem_duplication clears filters of reads: the same reads on every rank, from the replicated states)
		device_check(agpu_recover_internal_tandem_duplication(run.device, o.min_itd_support, o.min_itd_allele_fraction, &count)); run.note("recover_internal_tandem_duplication", count);
		device_check(agpu_filter_both_intronic(run.device, &count)); run.note("filter_both_intronic", count);
		if (o.known_fusions_file && run.enabled(F_known_fusions)) {
			const agpu_range_rule* rules = nullptr; uint32_t n_rules = 0;
			host_check(ahost_load_range_rules(run.host, o.known_fusions_file, 0, &rules, &n_rules));
			device_check(agpu_recover_known_fusions(run.device, rules, n_rules, max_mate_gap, &count)); run.note("recover_known_fusions", count);
		}
	});
	// :483-486 filter_in_vitro: chimeric fragments per gene -- sums over the ranks; the discordant mates clipped at a breakpoint -- counted per candidate over the reads a rank holds
	std::vector<int64_t> gene_reads; std::vector<uint8_t> clipped; uint64_t n_clipped = 0;
	together(run.ranks, [&] {
		std::vector<uint32_t> mine(n_genes > 0 ? n_genes : 1, 0);
		device_check(agpu_gene_read_counts(run.device, mine.data()));
		gene_reads.assign(mine.begin(), mine.end());
		device_check(agpu_in_vitro_clipped_mates(run.device, &n_clipped));
		clipped.resize((n_clipped > 0 ? n_clipped : 1) * AGPU_CLIPPED_MATES_ENTRY_BYTES);
		device_check(agpu_copy_in_vitro_clipped_mates(run.device, clipped.data()));
	});
	sum_over_ranks(run, gene_reads.data(), gene_reads.size(), "chimeric fragments per gene");
	gather_bytes(run, clipped.data(), n_clipped * AGPU_CLIPPED_MATES_ENTRY_BYTES, all, sizes, "clipped discordant mates of the candidates");
	uint64_t n_positions = 0;
	together(run.ranks, [&] {
		std::vector<uint32_t> sums(gene_reads.begin(), gene_reads.end());
		device_check(agpu_set_gene_read_counts(run.device, sums.data()));
		device_check(agpu_filter_in_vitro_sharded(run.device, o.high_expression_quantile, all.empty() ? nullptr : all.data(), all.size() / AGPU_CLIPPED_MATES_ENTRY_BYTES, &count)); run.note("filter_in_vitro", count);
		// :489-544
		device_check(agpu_recover_both_spliced(run.device, 200, 0.998f, 1000, 1000, &count)); run.note("recover_both_spliced", count);
		device_check(agpu_select_most_supported_breakpoints(run.device, &count)); run.note("select_most_supported_breakpoints", count);
		device_check(agpu_filter_marginal_read_through(run.device, &count)); run.note("filter_marginal_read_through", count);
		device_check(agpu_recover_many_spliced(run.device, o.min_spliced_events, &count)); run.note("recover_many_spliced", count);
		if (o.genomic_breakpoints_file && run.enabled(F_no_genomic_support)) {
			device_check(agpu_assign_confidence(run.device, nullptr));
			device_check(agpu_filter_no_genomic_support(run.device, &count)); run.note("filter_no_genomic_support", count);
		}
		if (o.blacklist_file && run.enabled(F_blacklist)) {
			const agpu_range_rule* rules = nullptr; uint32_t n_rules = 0;
			host_check(ahost_load_range_rules(run.host, o.blacklist_file, 1, &rules, &n_rules));
			device_check(agpu_filter_blacklisted_ranges(run.device, rules, n_rules, params.evalue_cutoff, max_mate_gap, &count)); run.note("filter_blacklisted_ranges", count);
		}
		device_check(agpu_filter_short_anchor(run.device, o.min_anchor_length, &count)); run.note("filter_short_anchor", count);
		device_check(agpu_filter_end_to_end(run.device, &count)); run.note("filter_end_to_end_fusions", count);
		device_check(agpu_filter_no_coverage(run.device, &count)); run.note("filter_no_coverage", count);
		// :546-560
		run.say("Indexing gene sequences ");
		device_check(agpu_make_kmer_index(run.device, (int32_t) ((float) max_mate_gap + 2.0f * read_length_mean), &n_positions));
		device_check(agpu_filter_homologs(run.device, o.max_homolog_identity, &count)); run.note("filter_homologs", count);
	});
	// :562-565 filter_mismappers: every rank re-aligns the reads it holds (the lists say which), the mis-mappers travel as states, the candidates are judged on every rank
	{ const double before = now_seconds();
	  together(run.ranks, [&] { device_check(agpu_filter_mismappers_search(run.device, max_mate_gap, &discarded_reads)); });
	  const double searched = now_seconds();
	  exchange_read_state(run);
	  run.exchange_verdicts = now_seconds() - searched;
	  together(run.ranks, [&] { device_check(agpu_filter_mismappers_judge(run.device, &count)); });
	  mismappers_seconds = now_seconds() - before; }
	run.note("filter_mismappers", count);
	together(run.ranks, [&] {
		if (o.genomic_breakpoints_file && run.enabled(F_genomic_support)) { device_check(agpu_recover_genomic_support(run.device, &count)); run.note("recover_genomic_support", count); }
		if ((o.genomic_breakpoints_file && run.enabled(F_genomic_support)) || run.enabled(F_many_spliced)) { device_check(agpu_select_most_supported_breakpoints(run.device, &count)); run.note("select_most_supported_breakpoints", count); }
		device_check(agpu_recover_isoforms(run.device, &count)); run.note("recover_isoforms", count);
		run.say("Assigning confidence scores to events ");
	});
}

// already_fed: the bytes of the file are in HBM (feed_file ran on the feeder thread of a submitted sample); sample_started: when the caller began to wait for this sample
void run_sample(Run& run, bool already_fed, double sample_started) {
	const arriba_workflow_options& o = run.options;
	if (!o.chimeric_bam_file || !o.output_file) throw Failure{ "ERROR: assembly, gene annotation, alignments and output file are required" };
	if (run.timing) memset(run.timing, 0, sizeof(*run.timing));
	agpu_params& params = run.params;
	run.exchange_parts = run.exchange_verdicts = run.exchange_rows = 0;
	if (run.device_ingest) {
		if (!already_fed) together(run.ranks, [&] { feed_file(run); });
		if (!(already_fed && run.ingest_finished_ahead)) finish_device_ingest(run, sample_started);
		note_device_ingest(run, sample_started);
	}
	else {
		if (run.ranks != nullptr) throw Failure{ "ERROR: one sample over several ranks needs read_chimeric_alignments on the device (host_ingest = 0): the parts of the batch are exchanged in device format" };
		host_check(ahost_ingest_bam_file(run.host, o.chimeric_bam_file, o.device.external_duplicate_marking, o.device.max_itd_length));
		device_check(agpu_upload_genome(run.device, ahost_genome_view(run.host)));
		device_check(agpu_upload_batch(run.device, ahost_batch_view(run.host)));
		run.n_fragments = ahost_batch_view(run.host)->n;
		if (run.timing) run.timing->feed = now_seconds() - sample_started;
	}
	run.mapped_reads = ahost_mapped_reads(run.host);
	run.note("read_chimeric_alignments", run.n_fragments); // (the log line follows; `note` prints the lines of the stages behind it)
	if (o.log_to_stdout) std::cout << Run::time_string() << " Reading chimeric alignments from '" << o.chimeric_bam_file << "' (total=" << run.n_fragments << ")" << std::endl;
	const double stages_started = now_seconds();
	double mismappers_seconds = 0;

	// (one sample over several ranks: every rank holds the whole batch by now and runs the stages up to filter_homologs on it -- identical inputs, identical kernels, nothing to
	// exchange; how the ranks fared is told once behind them, in front of the exchange of filter_mismappers)
	uint64_t count = 0, discarded_reads = 0; int32_t max_mate_gap = 0;
	if (run.sharded) run_stages_sharded(run, params, max_mate_gap, mismappers_seconds); // (the reads of the sample are sharded over the ranks: the same stages with their exchanges)
	else {
	together(run.ranks, [&] {
	// :141-325 multi-mappers, strandedness, annotation
	device_check(agpu_mark_multimappers(run.device, &count)); run.note("mark_multimappers", count);
	if (o.device.strandedness <= 2) params.strandedness = o.device.strandedness;
	else if (run.device_ingest) { int verdict = 0; device_check(agpu_detect_strandedness(run.device, &verdict)); params.strandedness = (uint8_t) verdict; }
	else params.strandedness = (uint8_t) ahost_detect_strandedness(run.host);
	if (o.device.strandedness > 2) run.say(std::string("Detecting strandedness (") + (params.strandedness == 1 ? "yes" : params.strandedness == 2 ? "reverse" : "no") + ")");
	if (params.strandedness != 0) run.say("Assigning strands to alignments ");
	run.say("Annotating alignments ");
	device_check(agpu_set_params(run.device, &params));
	device_check(agpu_annotate(run.device, &run.dummy_genes));
	// :327-350 duplicates, uninteresting and viral contigs (the per-contig verdicts are sequential host work)
	uint64_t n_pairs = 0;
	device_check(agpu_get_viral_integration_sites(run.device, nullptr, 0, &n_pairs));
	std::vector<uint32_t> pairs(2 * (n_pairs > 0 ? n_pairs : 1));
	device_check(agpu_get_viral_integration_sites(run.device, pairs.data(), n_pairs, &n_pairs));
	const uint32_t n_genes = ahost_annotation_view(run.host)->n_genes + run.dummy_genes, n_contigs = ahost_contig_count(run.host);
	std::vector<uint8_t> gene_bits(n_genes > 0 ? n_genes : 1), top(n_contigs > 0 ? n_contigs : 1), low(top.size());
	device_check(agpu_get_gene_table(run.device, 0, n_genes, nullptr, nullptr, nullptr, gene_bits.data(), nullptr));
	host_check(ahost_viral_verdicts(run.host, pairs.data(), n_pairs, gene_bits.data(), n_genes, o.top_viral_contigs, o.viral_contig_min_covered_fraction, top.data(), low.data()) < 0 ? -1 : 0);
	device_check(agpu_read_filters_stage1(run.device, top.data(), low.data()));
	// :352-364 fragment length
	std::vector<int32_t> mate_gaps(100001);
	uint32_t n_samples = 0; uint64_t visited = 0;
	device_check(agpu_fragment_length_samples(run.device, mate_gaps.data(), &n_samples, &visited));
	float mate_gap_mean = 0, mate_gap_stddev = 0, read_length_mean = 0;
	if (run.device_ingest) { // the sequential float sum of the read lengths (hazard H4) stays on the host, over the lengths of the fragments the reference's loop visits
		const uint64_t count = visited < run.n_fragments ? visited : run.n_fragments;
		std::vector<uint32_t> length1(count > 0 ? count : 1), length2(length1.size());
		device_check(agpu_get_read_lengths(run.device, 0, count, length1.data(), length2.data()));
		if (ahost_estimate_fragment_length_from_sums(mate_gaps.data(), n_samples, ahost_read_length_sum_of(0, length1.data(), length2.data(), count), count, params.fragment_length, &mate_gap_mean, &mate_gap_stddev, &read_length_mean, &max_mate_gap) < 0)
			throw Failure{ std::string("ERROR: ") + ahost_last_error() };
	} else if (ahost_estimate_fragment_length(run.host, mate_gaps.data(), n_samples, visited, params.fragment_length, &mate_gap_mean, &mate_gap_stddev, &read_length_mean, &max_mate_gap) < 0) throw Failure{ std::string("ERROR: ") + ahost_last_error() };
	// :366-409 the read-level filters
	std::vector<uint64_t> remaining(AGPU_FILTER_COUNT);
	device_check(agpu_read_filters_stage2(run.device, remaining.data()));
	std::ostringstream fragment_length_line;
	fragment_length_line << "Estimating fragment length ";
	if (n_samples >= 10000) fragment_length_line << "(mate gap mean=" << mate_gap_mean << ", mate gap stddev=" << mate_gap_stddev << ", read length mean=" << read_length_mean << ")";
	static const struct { unsigned id; const char* name; } read_filters[] = { { 1, "filter_duplicates" }, { 30, "filter_uninteresting_contigs" }, { 31, "filter_viral_contigs" }, { 32, "filter_top_expressed_viral_contigs" },
		{ 33, "filter_low_coverage_viral_contigs" }, { 4, "filter_proximal_read_through" }, { 2, "filter_inconsistently_clipped_mates" }, { 3, "filter_homopolymer" }, { 6, "filter_small_insert_size" }, { 7, "filter_long_gap" },
		{ 5, "filter_same_gene" }, { 8, "filter_hairpin" }, { 10, "filter_mismatches" }, { 36, "filter_low_entropy" } };
	for (size_t f = 0; f < sizeof(read_filters) / sizeof(read_filters[0]); ++f) {
		if (f == 5) run.say(fragment_length_line.str()); // the estimate sits between the contig filters and the rest (source/arriba.cpp:352)
		run.note(read_filters[f].name, remaining[read_filters[f].id]);
	}

	// :411-460 candidates
	device_check(agpu_find_fusions(run.device, max_mate_gap, &count));
	run.n_candidates = count;
	if (o.log_to_stdout) { // the reference's "(total=N)" counts the candidates with at least one read that no filter discarded (source/fusions.cpp:467-472)
		std::vector<uint8_t> filter(count > 0 ? count : 1);
		device_check(agpu_get_candidates(run.device, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, filter.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
		uint64_t unfiltered = 0;
		for (uint64_t c = 0; c < count; ++c) unfiltered += filter[c] == 0;
		run.progress("find_fusions", unfiltered);
	}
	if (run.report && run.report->n_stages < sizeof(run.report->stages) / sizeof(run.report->stages[0])) { arriba_workflow_stage& entry = run.report->stages[run.report->n_stages++]; snprintf(entry.stage, sizeof(entry.stage), "find_fusions"); entry.count = count; }
	if (!run.device_ingest) device_check(agpu_upload_coverage(run.device, ahost_coverage_view(run.host))); // (the device ingest built coverage_t in place)
	if (o.genomic_breakpoints_file) {
		const agpu_genomic_breakpoint* variants = nullptr; uint32_t n_variants = 0;
		host_check(ahost_load_genomic_breakpoints(run.host, o.genomic_breakpoints_file, &variants, &n_variants));
		device_check(agpu_mark_genomic_support(run.device, variants, n_variants, o.max_genomic_breakpoint_distance, &count)); run.note("mark_genomic_support", count);
	}
	device_check(agpu_merge_adjacent_fusions(run.device, 5, &count)); run.note("merge_adjacent_fusions", count);
	uint64_t discarded[3];
	device_check(agpu_filter_multimappers(run.device, &count, &discarded_reads)); run.note("filter_multimappers", count);
	device_check(agpu_candidate_iteration_order(run.device, nullptr)); // hazard H2: the order in which the reference's container is walked, kept on the device
	run.say("Estimating expected number of fusions by random chance (e-value) ");
	device_check(agpu_estimate_expected_fusions(run.device, run.mapped_reads, nullptr));
	device_check(agpu_filter_candidate_predicates(run.device, discarded));
	if (o.log_to_stdout) { // the three predicates run in one kernel; their counts follow from what each discarded
		const uint64_t before = count; // unfiltered candidates behind filter_multimappers
		run.progress("filter_non_coding_neighbors", before - discarded[0]); run.progress("filter_intragenic_both_exonic", before - discarded[0] - discarded[1]); run.progress("filter_min_support", before - discarded[0] - discarded[1] - discarded[2]);
	}
	device_check(agpu_filter_relative_support(run.device, &count)); run.note("filter_relative_support", count);
	// :463-544 the candidate-level filters (each stage skips itself when its filter is switched off with -f)
	device_check(agpu_recover_internal_tandem_duplication(run.device, o.min_itd_support, o.min_itd_allele_fraction, &count)); run.note("recover_internal_tandem_duplication", count);
	device_check(agpu_filter_both_intronic(run.device, &count)); run.note("filter_both_intronic", count);
	if (o.known_fusions_file && run.enabled(F_known_fusions)) {
		const agpu_range_rule* rules = nullptr; uint32_t n_rules = 0;
		host_check(ahost_load_range_rules(run.host, o.known_fusions_file, 0, &rules, &n_rules));
		device_check(agpu_recover_known_fusions(run.device, rules, n_rules, max_mate_gap, &count)); run.note("recover_known_fusions", count);
	}
	device_check(agpu_filter_in_vitro(run.device, o.high_expression_quantile, &count)); run.note("filter_in_vitro", count);
	device_check(agpu_recover_both_spliced(run.device, 200, 0.998f, 1000, 1000, &count)); run.note("recover_both_spliced", count); // the constants of source/arriba.cpp:491
	device_check(agpu_select_most_supported_breakpoints(run.device, &count)); run.note("select_most_supported_breakpoints", count);
	device_check(agpu_filter_marginal_read_through(run.device, &count)); run.note("filter_marginal_read_through", count);
	device_check(agpu_recover_many_spliced(run.device, o.min_spliced_events, &count)); run.note("recover_many_spliced", count);
	if (o.genomic_breakpoints_file && run.enabled(F_no_genomic_support)) {
		device_check(agpu_assign_confidence(run.device, nullptr)); // this filter looks at the confidence (:516-523)
		device_check(agpu_filter_no_genomic_support(run.device, &count)); run.note("filter_no_genomic_support", count);
	}
	if (o.blacklist_file && run.enabled(F_blacklist)) {
		const agpu_range_rule* rules = nullptr; uint32_t n_rules = 0;
		host_check(ahost_load_range_rules(run.host, o.blacklist_file, 1, &rules, &n_rules));
		device_check(agpu_filter_blacklisted_ranges(run.device, rules, n_rules, params.evalue_cutoff, max_mate_gap, &count)); run.note("filter_blacklisted_ranges", count);
	}
	device_check(agpu_filter_short_anchor(run.device, o.min_anchor_length, &count)); run.note("filter_short_anchor", count);
	device_check(agpu_filter_end_to_end(run.device, &count)); run.note("filter_end_to_end_fusions", count);
	device_check(agpu_filter_no_coverage(run.device, &count)); run.note("filter_no_coverage", count);
	// :546-565 the k-mer index and the two filters that use it; padding as in :552 (float arithmetic, truncated)
	uint64_t n_positions = 0;
	run.say("Indexing gene sequences ");
	device_check(agpu_make_kmer_index(run.device, (int32_t) ((float) max_mate_gap + 2.0f * read_length_mean), &n_positions));
	device_check(agpu_filter_homologs(run.device, o.max_homolog_identity, &count)); run.note("filter_homologs", count);
	});
	{ const double before = now_seconds();
	  if (run.ranks != nullptr) filter_mismappers_over_ranks(run, max_mate_gap, &count, &discarded_reads);
	  else device_check(agpu_filter_mismappers(run.device, max_mate_gap, &count, &discarded_reads));
	  mismappers_seconds = now_seconds() - before; }
	run.note("filter_mismappers", count);
	together(run.ranks, [&] {
	// :567-584
	if (o.genomic_breakpoints_file && run.enabled(F_genomic_support)) { device_check(agpu_recover_genomic_support(run.device, &count)); run.note("recover_genomic_support", count); }
	if ((o.genomic_breakpoints_file && run.enabled(F_genomic_support)) || run.enabled(F_many_spliced)) { device_check(agpu_select_most_supported_breakpoints(run.device, &count)); run.note("select_most_supported_breakpoints", count); }
	device_check(agpu_recover_isoforms(run.device, &count)); run.note("recover_isoforms", count);
	run.say("Assigning confidence scores to events ");
	});
	}
	const double output_started = now_seconds();
	write_output_files(run, max_mate_gap);
	if (run.timing) {
		const double finished = now_seconds();
		run.timing->filter_mismappers = mismappers_seconds; run.timing->stages = output_started - stages_started - mismappers_seconds;
		run.timing->output = finished - output_started; run.timing->total = finished - sample_started;
		run.timing->exchange_parts = run.exchange_parts; run.timing->exchange_verdicts = run.exchange_verdicts; run.timing->exchange_rows = run.exchange_rows;
		run.timing->shard_fragments = run.sharded ? (double) run.local_fragments : 0; run.timing->exchanged_bytes = (double) run.exchanged_bytes;
	}
}

}

extern "C" {

void arriba_workflow_default_options(arriba_workflow_options* options) {
	memset(options, 0, sizeof(*options));
	agpu_default_params(&options->device);
	options->device.strandedness = 3; // -s auto
	options->min_itd_support = 10; options->min_itd_allele_fraction = 0.07f; options->high_expression_quantile = 0.998f; options->min_spliced_events = 4; options->min_anchor_length = 23;
	options->max_homolog_identity = 0.3f; options->top_viral_contigs = 5; options->viral_contig_min_covered_fraction = 0.05f; options->max_genomic_breakpoint_distance = 100000;
}

const char* arriba_workflow_last_error(void) { return g_error.c_str(); }

int arriba_workflow_run(const arriba_workflow_options* options, arriba_workflow_report* report) {
	if (!options) { g_error = "ERROR: null options"; return -1; }
	if (report) report->n_stages = 0;
	try {
		if (!options->assembly_file || !options->gene_annotation_file || !options->chimeric_bam_file || !options->output_file) throw Failure{ "ERROR: assembly, gene annotation, alignments and output file are required" };
		Run run(*options);
		run.report = report;
		open_session(run);
		prepare_sample(run);
		run_sample(run, false, now_seconds());
		return 0;
	}
	catch (const Failure& failure) { g_error = failure.text; return -1; }
	catch (const std::exception& e) { g_error = std::string("ERROR: ") + e.what(); return -1; }
}

// A resident session.  One lane = one Run (host session + device context); a second lane (same files, a sibling context that shares the scratch buffers of the first:
// agpu_create_sibling) is made at the first arriba_workflow_submit.  Submitted samples are fed in the order they were submitted, each on a thread of its own, into the lane
// that is not at work; the stream and the tables of the ingest exist once, so a feed starts when the ingest of the sample before has finished (agpu_ingest_finish) -- which
// is when the stages of that sample start.  arriba_workflow_sample takes the oldest submitted sample (or submits the one it is given).
struct arriba_workflow_session {
	Run* lanes[2];
	int processed_lane = 0; // of the sample arriba_workflow_sample worked on last
	struct Submitted { std::string bam; int lane = 0; std::thread feeder; bool fed = false, ingest_finished = false, started = false; std::string error; int error_code = 0; };
	std::deque<std::unique_ptr<Submitted>> queue; // oldest first; at most two
	std::mutex mutex; std::condition_variable changed;
	bool ingest_busy = false; // a lane is between agpu_ingest_begin and agpu_ingest_finish
	bool defer_output = false;
	bool retrying = false; // arriba_workflow_sample runs a sample again after the device ran out of memory with two lanes (below)
	bool finish_ahead = false; // arriba_workflow_finish_ahead: the feeder of a sample also finishes its ingest (the lanes keep their batch buffers)
	// arriba_workflow_set_communicator / arriba_workflow_join_rccl: one sample over the ranks of a job; `joined`: the RCCL communicator this session made itself (and the state of
	// its callbacks: host bytes bounced through the device of lane 0)
	arriba_workflow_communicator communicator; bool over_ranks = false;
	struct Joined { agpu_ctx* device = nullptr; void* comm = nullptr; uint32_t size = 0; } joined;
	void leave() { if (joined.comm != nullptr) { agpu_rccl_leave(joined.comm); joined.comm = nullptr; } over_ranks = false; for (int k = 0; k < 2; ++k) if (lanes[k]) lanes[k]->ranks = nullptr; }
	std::string deferred_error; // of a writer that was joined on the way (reported by the next arriba_workflow_sample / arriba_workflow_flush)
	double deferred_seconds = 0; // the writer joined last
	std::mutex writer_mutex; // (a writer is joined by the thread that calls the session or by the feeder of the lane's next sample)
	void join_writer_of(int lane) {
		std::lock_guard<std::mutex> lock(writer_mutex);
		if (lanes[lane] == nullptr || !lanes[lane]->writer.joinable()) return;
		lanes[lane]->writer.join();
		deferred_seconds = lanes[lane]->writer_seconds;
		if (!lanes[lane]->writer_error.empty() && deferred_error.empty()) deferred_error = lanes[lane]->writer_error;
		lanes[lane]->writer_error.clear();
	}
	std::string take_deferred_error() { std::lock_guard<std::mutex> lock(writer_mutex); std::string text; text.swap(deferred_error); return text; }
	arriba_workflow_session(const arriba_workflow_options& o) { lanes[0] = new Run(o); lanes[1] = nullptr; }
	~arriba_workflow_session() {
		drain();
		leave();
		delete lanes[1]; delete lanes[0]; // (the sibling first: either order is allowed)
	}
	// the feeds in flight are waited for and thrown away
	void drain() {
		while (!queue.empty()) {
			Submitted& sample = *queue.front();
			if (sample.feeder.joinable()) sample.feeder.join();
			Run& run = *lanes[sample.lane];
			abandon(sample, run);
			{ std::lock_guard<std::mutex> lock(mutex); queue.pop_front(); }
		}
	}
	// a sample that was fed (or began to be) and will not be finished: the next agpu_ingest_begin of its lane starts over, the stream and the tables are free
	void abandon(Submitted& sample, Run& run) {
		if (run.bam_open) { ahost_bam_close(run.host); run.bam_open = false; }
		if (sample.started && !sample.ingest_finished) { { std::lock_guard<std::mutex> lock(mutex); sample.ingest_finished = true; } release_ingest(); }
	}
	void release_ingest() { { std::lock_guard<std::mutex> lock(mutex); ingest_busy = false; } changed.notify_all(); }
	void submit(const char* bam) {
		if (queue.size() >= 2) throw Failure{ "ERROR: two samples are submitted already (one at work, one being fed): call arriba_workflow_sample first" };
		int lane = 0;
		if (!queue.empty()) lane = 1 - queue.back()->lane; // (the lane that is not at work)
		else if (lanes[1] != nullptr) lane = 1 - processed_lane;
		if (lane == 1 && lanes[1] == nullptr) { // the second lane: the same reference data again (the host session holds the state of a sample next to them), a sibling context
			std::unique_ptr<Run> second(new Run(lanes[0]->options));
			second->options.log_to_stdout = 0; // (its loading is not a step of any sample's log)
			const arriba_workflow_options& o = second->options;
			second->host = ahost_open(o.assembly_file, o.gene_annotation_file, o.interesting_contigs, o.viral_contigs, o.gtf_features);
			if (!second->host) throw Failure{ std::string("ERROR: ") + ahost_last_error() };
			second->params = lanes[0]->params;
			second->device = agpu_create_sibling(lanes[0]->device);
			if (!second->device) throw Failure{ std::string("ERROR: ") + agpu_last_error() };
			device_check(agpu_upload_annotation(second->device, ahost_annotation_view(second->host)));
			second->options.log_to_stdout = lanes[0]->options.log_to_stdout;
			second->ranks = lanes[0]->ranks;
			lanes[1] = second.release();
		}
		Run& run = *lanes[lane];
		run.bam_path = bam; run.options.chimeric_bam_file = run.bam_path.c_str();
		run.timing = nullptr; run.report = nullptr;
		prepare_sample(run);
		std::unique_ptr<Submitted> sample(new Submitted());
		sample->bam = bam; sample->lane = lane;
		Submitted* mine = sample.get();
		{ std::lock_guard<std::mutex> lock(mutex); queue.push_back(std::move(sample)); }
		if (!run.device_ingest) { join_writer_of(lane); std::lock_guard<std::mutex> lock(mutex); mine->fed = true; return; } // (the host ingest reads the file inside arriba_workflow_sample)
		const bool ahead = queue.size() > 1; // fed beside the stages of the sample in front of it: the threads that read the file leave processors to the thread that runs those
		mine->feeder = std::thread([this, mine, &run, lane, ahead] {
			if (ahead) ahost_limit_threads_of_this_thread(std::max(2u, ahost_cpu_budget() / 2));
			{ std::unique_lock<std::mutex> lock(mutex); changed.wait(lock, [&] { return !ingest_busy && (queue.front().get() == mine || queue.front()->ingest_finished); }); ingest_busy = true; mine->started = true; }
			run.ingest_finished_ahead = false;
			try {
				feed_file(run);
				if (finish_ahead && !over_ranks) { // what is left of read_chimeric_alignments behind the last piece, here and now: the stages of the sample in front may still run on the other lane
					finish_device_ingest(run, now_seconds());
					run.ingest_finished_ahead = true;
					{ std::lock_guard<std::mutex> lock(mutex); mine->ingest_finished = true; }
					release_ingest();
				}
			}
			catch (const Failure& failure) { mine->error = failure.text; }
			catch (const std::exception& e) { mine->error = std::string("ERROR: ") + e.what(); }
			{ std::lock_guard<std::mutex> lock(mutex); mine->fed = true; }
			changed.notify_all();
		});
	}
};

arriba_workflow_session* arriba_workflow_open(const arriba_workflow_options* options) {
	if (!options) { g_error = "ERROR: null options"; return nullptr; }
	arriba_workflow_session* session = nullptr;
	try { session = new arriba_workflow_session(*options); open_session(*session->lanes[0]); return session; }
	catch (const Failure& failure) { g_error = failure.text; }
	catch (const std::exception& e) { g_error = std::string("ERROR: ") + e.what(); }
	delete session;
	return nullptr;
}

int arriba_workflow_submit(arriba_workflow_session* session, const char* chimeric_bam_file) {
	if (!session || !chimeric_bam_file) { g_error = "ERROR: null argument"; return -1; }
	try { session->submit(chimeric_bam_file); return 0; }
	catch (const Failure& failure) { g_error = failure.text; }
	catch (const std::exception& e) { g_error = std::string("ERROR: ") + e.what(); }
	return -1;
}

int arriba_workflow_sample(arriba_workflow_session* session, const char* chimeric_bam_file, const char* output_file, const char* discarded_output_file, arriba_workflow_report* report, arriba_workflow_timing* timing) {
	if (!session || !chimeric_bam_file || !output_file) { g_error = "ERROR: assembly, gene annotation, alignments and output file are required"; return -1; }
	if (report) report->n_stages = 0;
	int status = 0;
	Run* lane = nullptr;
	try {
		const double sample_started = now_seconds();
		// (over several ranks: what can fail in front of the first collective of the sample -- the second lane of the session that cannot be made, a sample asked for out of order -- is
		//  told to the other ranks like every failure behind it, so that nobody waits in the status exchange of a rank that has left the call: advisor, round 5)
		together(session->over_ranks ? &session->communicator : nullptr, [&] {
			if (session->queue.empty()) session->submit(chimeric_bam_file);
			if (session->queue.front()->bam != chimeric_bam_file) throw Failure{ "ERROR: samples are worked on in the order they were submitted: '" + session->queue.front()->bam + "' comes first" };
		});
		arriba_workflow_session::Submitted& sample = *session->queue.front();
		Run& run = *session->lanes[sample.lane];
		lane = &run;
		run.output_path = output_file; run.discarded_path = discarded_output_file ? discarded_output_file : "";
		run.options.output_file = run.output_path.c_str(); run.options.discarded_output_file = discarded_output_file ? run.discarded_path.c_str() : nullptr;
		run.report = report; run.timing = timing;
		run.defer_output = session->defer_output && !session->over_ranks;
		const int other = 1 - sample.lane;
		run.before_host_writer = [session, other] { session->join_writer_of(other); };
		// (advisor, round 4: nothing may throw between here and `done` -- the clean-up behind the catch blocks clears the lane's options, which the feeder of this sample reads
		// until it is joined, and a sample that stays at the front of the queue with cleared options cannot be retried)
		if (sample.feeder.joinable()) sample.feeder.join(); // (the feed of this sample: under the stages of the sample before if it was submitted ahead)
		session->join_writer_of(sample.lane); // (the last file of the lane's sample before: written from a detached sample beside this sample's feed; done long ago, normally)
		struct Done { arriba_workflow_session& session; ~Done() { // on every way out: the ingest buffers are free for the next feed, the sample leaves the queue
			arriba_workflow_session::Submitted& sample = *session.queue.front();
			session.abandon(sample, *session.lanes[sample.lane]); // (nothing to do behind a sample that went through)
			{ std::lock_guard<std::mutex> lock(session.mutex); session.queue.pop_front(); }
			session.changed.notify_all(); } } done = { *session };
		together(run.ranks, [&] { if (!sample.error.empty()) throw Failure{ sample.error, sample.error_code }; }); // (the feed of this rank's part)
		run.after_ingest = [session, &sample] { { std::lock_guard<std::mutex> lock(session->mutex); sample.ingest_finished = true; } session->release_ingest(); };
		run_sample(run, run.device_ingest, sample_started);
		// an I/O error on the deferred file of an EARLIER sample is reported behind this one: this sample has gone through and its own files are as they should be (its last
		// one possibly still being written, arriba_workflow_flush); the call fails with the earlier sample's message so that the caller hears of it at the first call after it happened
		{ const std::string text = session->take_deferred_error(); if (!text.empty()) throw Failure{ text + " (writing the last file of an earlier sample; the files of this sample are not affected)" }; }
	}
	catch (const Failure& failure) { g_error = failure.text; g_error_code = failure.code; status = -1; }
	catch (const std::exception& e) { g_error = std::string("ERROR: ") + e.what(); g_error_code = 0; status = -1; }
	if (lane) { lane->after_ingest = nullptr; lane->before_host_writer = nullptr; lane->options.chimeric_bam_file = nullptr; lane->options.output_file = nullptr; lane->options.discarded_output_file = nullptr; lane->report = nullptr; lane->timing = nullptr; session->processed_lane = (int) (lane == session->lanes[1]); }
	// The device ran out of memory while the session had two lanes (advisor, round 4): their contexts share one pool of scratch buffers, of which nothing is idle while one lane
	// feeds and the other runs its stages, so the device library gives nothing back by itself (DeviceBuffer::release_idle_buffers).  The session does what INTEGRATION.md ("Memory")
	// used to ask of the caller: what was fed ahead is thrown away, the second lane is closed -- the pool belongs to one context again, which gives back what it keeps for its next
	// sample when an allocation fails --, and the sample is run again, alone.  Once; the sample that was fed ahead is submitted again behind it.
	if (status != 0 && lane != nullptr && !session->retrying && !session->over_ranks && session->lanes[1] != nullptr && g_error_code == AGPU_ERR_NO_MEMORY) { // (the status code of the device library, not a word of its message: review of round 5)
		const std::string first_error = g_error;
		const std::string behind = session->queue.empty() ? std::string() : session->queue.front()->bam;
		session->drain();
		session->join_writer_of(0); session->join_writer_of(1);
		// (an I/O error on the deferred file of an EARLIER sample stays noted -- advisor, round 5: it was dropped here, and the caller never learnt that the file is incomplete -- and is
		//  reported behind the sample that is run again, as it is behind any sample: take_deferred_error in the call below)
		if (lane == session->lanes[1]) std::swap(session->lanes[0], session->lanes[1]);
		delete session->lanes[1]; session->lanes[1] = nullptr; session->processed_lane = 0;
		fprintf(stderr, "arriba_workflow_sample: %s -- with two samples in flight; '%s' is run again with the device to itself\n", first_error.c_str(), chimeric_bam_file);
		session->retrying = true;
		status = arriba_workflow_sample(session, chimeric_bam_file, output_file, discarded_output_file, report, timing);
		session->retrying = false;
		if (status == 0 && !behind.empty()) { // (as its caller submitted it: a failure to feed it is reported by the call that asks for it)
			try { session->submit(behind.c_str()); }
			catch (const Failure&) {} catch (const std::exception&) {}
		}
	}
	return status;
}

int arriba_workflow_defer_output(arriba_workflow_session* session, int on) {
	if (!session) { g_error = "ERROR: null argument"; return -1; }
	session->defer_output = on != 0;
	return 0;
}
int arriba_workflow_finish_ahead(arriba_workflow_session* session, int on) {
	if (!session) { g_error = "ERROR: null argument"; return -1; }
	if (!session->queue.empty()) { g_error = "ERROR: arriba_workflow_finish_ahead: samples are submitted already"; return -1; }
	if (session->lanes[0]->options.host_ingest) { if (!on) return 0; g_error = "ERROR: arriba_workflow_finish_ahead needs read_chimeric_alignments on the device"; return -1; }
	if (agpu_keep_batch_buffers(session->lanes[0]->device, on) != 0) { g_error = std::string("ERROR: ") + agpu_last_error(); return -1; } // (both lanes, also the one made later: agpu_create_sibling copies the flag)
	session->finish_ahead = on != 0;
	return 0;
}
int arriba_workflow_set_communicator(arriba_workflow_session* session, const arriba_workflow_communicator* communicator) {
	if (!session) { g_error = "ERROR: null argument"; return -1; }
	if (!session->queue.empty()) { g_error = "ERROR: arriba_workflow_set_communicator: samples are submitted already"; return -1; }
	session->join_writer_of(0); session->join_writer_of(1);
	session->leave();
	if (communicator == nullptr) return 0;
	if (communicator->size == 0 || communicator->rank >= communicator->size || !communicator->all_gather || !communicator->all_reduce_int64 || !communicator->all_reduce_max_bytes) { g_error = "ERROR: arriba_workflow_set_communicator: rank, size and the three collectives are required"; return -1; }
	if (session->lanes[0]->options.host_ingest) { g_error = "ERROR: one sample over several ranks needs read_chimeric_alignments on the device (host_ingest = 0)"; return -1; }
	session->communicator = *communicator;
	session->over_ranks = true;
	for (int k = 0; k < 2; ++k) if (session->lanes[k]) session->lanes[k]->ranks = &session->communicator;
	return 0;
}
int arriba_workflow_rccl_unique_id(unsigned char* id) {
	if (agpu_rccl_unique_id(id) != AGPU_OK) { g_error = std::string("ERROR: ") + agpu_last_error(); return -1; }
	return 0;
}
int arriba_workflow_join_rccl(arriba_workflow_session* session, const unsigned char* id, uint32_t rank, uint32_t size) {
	if (!session || !id) { g_error = "ERROR: null argument"; return -1; }
	if (!session->queue.empty()) { g_error = "ERROR: arriba_workflow_join_rccl: samples are submitted already"; return -1; }
	session->leave();
	arriba_workflow_session::Joined& joined = session->joined;
	joined.device = session->lanes[0]->device; joined.size = size;
	if (agpu_rccl_join(joined.device, id, rank, size, &joined.comm) != AGPU_OK) { g_error = std::string("ERROR: ") + agpu_last_error(); joined.comm = nullptr; return -1; }
	typedef arriba_workflow_session::Joined Joined;
	arriba_workflow_communicator communicator;
	communicator.rank = rank; communicator.size = size; communicator.state = &joined; communicator.rccl_communicator = joined.comm;
	communicator.all_gather = [](void* state, const void* mine, void* all, uint64_t bytes) -> int { Joined& j = *(Joined*) state; return agpu_rccl_all_gather_host(j.device, j.comm, j.size, mine, all, bytes); };
	communicator.all_reduce_int64 = [](void* state, int64_t* values, uint64_t count, int operation) -> int { Joined& j = *(Joined*) state;
		return agpu_rccl_all_reduce_host(j.device, j.comm, values, count, operation == ARRIBA_WORKFLOW_MIN ? AGPU_REDUCE_MIN_INT64 : operation == ARRIBA_WORKFLOW_SUM ? AGPU_REDUCE_SUM_INT64 : AGPU_REDUCE_MAX_INT64); };
	communicator.all_reduce_max_bytes = [](void* state, uint8_t* values, uint64_t count) -> int { Joined& j = *(Joined*) state; return agpu_rccl_all_reduce_host(j.device, j.comm, values, count, AGPU_REDUCE_MAX_BYTES); };
	void* comm = joined.comm; joined.comm = nullptr; // (set_communicator starts with leave())
	const int status = arriba_workflow_set_communicator(session, &communicator);
	joined.comm = comm;
	if (status != 0) { agpu_rccl_leave(comm); joined.comm = nullptr; }
	return status;
}
int arriba_workflow_flush(arriba_workflow_session* session, double* seconds_of_last_writer) {
	if (!session) { g_error = "ERROR: null argument"; return -1; }
	session->join_writer_of(0); session->join_writer_of(1);
	if (seconds_of_last_writer) *seconds_of_last_writer = session->deferred_seconds;
	{ const std::string text = session->take_deferred_error(); if (!text.empty()) { g_error = text; return -1; } }
	return 0;
}
int arriba_workflow_cancel(arriba_workflow_session* session) {
	if (!session) { g_error = "ERROR: null argument"; return -1; }
	session->drain();
	return 0;
}

agpu_ctx* arriba_workflow_device(arriba_workflow_session* session) { return session ? session->lanes[session->processed_lane]->device : nullptr; }
agpu_ctx* arriba_workflow_lane_device(arriba_workflow_session* session, int lane) { return session && lane >= 0 && lane < 2 && session->lanes[lane] ? session->lanes[lane]->device : nullptr; }
ahost_session* arriba_workflow_host(arriba_workflow_session* session) { return session ? session->lanes[session->processed_lane]->host : nullptr; }
void arriba_workflow_close(arriba_workflow_session* session) { delete session; }

}
