// arriba_amd/csrc/host/output.cpp -- the output files (reference: source/output_fusions.cpp:468-710, :1043-1261): one line per candidate, the
// surviving ones sorted by support with the events of one gene pair kept together, the discarded ones in the iteration order of the
// reference's fusions_t (hazard H2; the rank comes from the device).  Runs on the host over the candidate table the device hands back:
// string formatting of a few thousand (fusions.tsv) to a few hundred thousand (discarded.tsv) rows.
#include "arriba_host.h"
#include "output.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include "transcript.h"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <fcntl.h>
#include <unistd.h>
#include <cstdlib>
#include <map>
#include <set>
#include <stdexcept>
#include <unordered_map>

namespace arriba {

namespace {

const unsigned N_FILTER_NAMES = FILTER_COUNT; // FILTER_NAMES: arriba_host.h (source/common.hpp:29-67)

// The threads that format the rows stay between the files of a resident service: a thread that is created per file starts with an empty allocator arena and
// grows it under the eyes of all others (page faults, mprotect), file after file; the threads of the pool keep theirs, and the pages of their pileups (transcript.cpp).
class FormatterPool {
public:
	~FormatterPool() { { std::lock_guard<std::mutex> lock(mutex_); stop_ = true; } wake_.notify_all(); for (size_t t = 0; t < threads_.size(); ++t) threads_[t].join(); }
	// `work` on n threads of the pool at once (it takes its items from a shared counter); returns when all of them have left it
	void run(unsigned n, const std::function<void()>& work) {
		std::unique_lock<std::mutex> lock(mutex_);
		while (threads_.size() < n) threads_.push_back(std::thread(&FormatterPool::serve, this, threads_.size()));
		work_ = &work; wanted_ = n; running_ = n; ++generation_;
		wake_.notify_all();
		done_.wait(lock, [&] { return running_ == 0; });
		work_ = NULL;
	}
private:
	void serve(size_t index) {
		worker_thread_starts();
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lock(mutex_);
		while (true) {
			wake_.wait(lock, [&] { return stop_ || (generation_ != seen && index < wanted_); });
			if (stop_) return;
			seen = generation_;
			const std::function<void()>* work = work_;
			lock.unlock();
			(*work)();
			lock.lock();
			if (--running_ == 0) done_.notify_all();
		}
	}
	std::mutex mutex_; std::condition_variable wake_, done_;
	std::vector<std::thread> threads_;
	const std::function<void()>* work_ = NULL;
	unsigned wanted_ = 0, running_ = 0; uint64_t generation_ = 0; bool stop_ = false;
};
FormatterPool& formatter_pool() { static FormatterPool pool; return pool; }
std::mutex formatter_pool_in_use; // (one file at a time)

struct Fusion { // one row, as the reference's fusion_t sees it
	uint32_t candidate;
	int gene1, gene2; contig_t contig1, contig2; position_t breakpoint1, breakpoint2;
	bool upstream1, upstream2, exonic1, exonic2, spliced1, spliced2, predicted_strand1, predicted_strand2, strands_ambiguous, transcript_start_gene1;
	bool transcript_start_ambiguous;
	unsigned split_reads1, split_reads2, discordant_mates; uint8_t filter, confidence; float evalue;
	unsigned supporting_reads() const { return split_reads1 + split_reads2 + discordant_mates; }
	bool is_read_through() const { return contig1 == contig2 && breakpoint2 - breakpoint1 < 400000 && !upstream1 && upstream2; } // source/common.hpp:265-269
};

// reference: sort_fusions_by_support (:468-484)
bool more_support(const Fusion& x, const Fusion& y) {
	if (x.confidence != y.confidence) return x.confidence > y.confidence;
	if (x.supporting_reads() != y.supporting_reads()) return x.supporting_reads() > y.supporting_reads();
	if (x.evalue != y.evalue) return x.evalue < y.evalue;
	if (x.gene1 != y.gene1) return x.gene1 < y.gene1; // gene ids: deterministic order, events of one gene pair together
	if (x.gene2 != y.gene2) return x.gene2 < y.gene2;
	if (x.breakpoint1 != y.breakpoint1) return x.breakpoint1 < y.breakpoint1;
	return x.breakpoint2 < y.breakpoint2;
}

struct Writer {
	const Annotation& annotation; const Contigs& contigs; const Coverage& coverage; const FusionTable& table;
	std::vector<GeneRecord> genes;  // GTF genes + the dummy genes of this sample
	FlatIndex gene_index;           // over all of them (the reference regenerates its index after adding the dummy genes, source/arriba.cpp:262-263)
	const FlatIndex& exon_index;

	// reference: gene_to_name (:498-545)
	std::string gene_to_name(int gene, contig_t contig, position_t breakpoint) const {
		if (!genes[gene].is_dummy) return genes[gene].name;
		std::string result;
		if ((size_t) contig >= gene_index.n_contigs()) return ".";
		const uint32_t begin = gene_index.contig_begin(contig), end = gene_index.contig_end(contig);
		const uint32_t hit2 = gene_index.lower_bound(contig, breakpoint);
		auto bucket_is_flanking = [&](uint32_t k) { // not empty, and its first gene (the lowest id) is not a dummy gene
			return gene_index.member_offset[k] < gene_index.member_offset[k + 1] && !genes[gene_index.members[gene_index.member_offset[k]]].is_dummy;
		};
		int64_t up = (int64_t) hit2 - 1; // the reverse iterator starts in front of the hit
		while (up >= (int64_t) begin && !bucket_is_flanking((uint32_t) up)) --up;
		if (up >= (int64_t) begin)
			for (uint32_t m = gene_index.member_offset[up]; m < gene_index.member_offset[up + 1]; ++m) {
				const GeneRecord& flanking = genes[gene_index.members[m]];
				if (flanking.is_dummy) continue;
				if (!result.empty()) result += ",";
				result += flanking.name + "(" + std::to_string((long long) (breakpoint - flanking.end)) + ")";
			}
		uint32_t down = hit2;
		while (down < end && !bucket_is_flanking(down)) ++down;
		if (down < end)
			for (uint32_t m = gene_index.member_offset[down]; m < gene_index.member_offset[down + 1]; ++m) {
				const GeneRecord& flanking = genes[gene_index.members[m]];
				if (flanking.is_dummy) continue;
				if (!result.empty()) result += ",";
				result += flanking.name + "(" + std::to_string((long long) (flanking.start - breakpoint)) + ")";
			}
		return result.empty() ? "." : result;
	}

	// reference: get_fusion_type (:547-614)
	std::string fusion_type(const Fusion& f, unsigned max_itd_length) const {
		const GeneRecord& gene1 = genes[f.gene1]; const GeneRecord& gene2 = genes[f.gene2];
		const bool any_dummy = gene1.is_dummy || gene2.is_dummy;
		if (f.contig1 != f.contig2) {
			if (any_dummy || (f.upstream1 == f.upstream2 && gene1.strand != gene2.strand) || (f.upstream1 != f.upstream2 && gene1.strand == gene2.strand)) return "translocation";
			if (((f.upstream1 && gene1.strand) || (!f.upstream1 && !gene1.strand)) && ((f.upstream2 && gene2.strand) || (!f.upstream2 && !gene2.strand))) return "translocation/3'-3'";
			return "translocation/5'-5'";
		}
		if (!f.upstream1 && f.upstream2) {
			const std::string kind = f.is_read_through() ? "deletion/read-through" : "deletion";
			if (any_dummy || gene1.strand == gene2.strand) return kind;
			return kind + ((gene1.strand || !gene2.strand) ? "/5'-5'" : "/3'-3'");
		}
		if (f.upstream1 == f.upstream2) {
			if (any_dummy || gene1.strand != gene2.strand) return "inversion";
			return (f.upstream1 && !gene1.strand) ? "inversion/5'-5'" : "inversion/3'-3'";
		}
		// upstream / downstream
		if (any_dummy || gene1.strand == gene2.strand) {
			if (f.gene1 == f.gene2 && f.spliced1 && f.spliced2) return "duplication/non-canonical_splicing";
			if (f.gene1 == f.gene2 && (unsigned) f.breakpoint2 - (unsigned) f.breakpoint1 < max_itd_length) return "duplication/ITD"; // is_internal_tandem_duplication (source/common.hpp:270-274); the directions are given here
			return "duplication";
		}
		return !gene1.strand ? "duplication/5'-5'" : "duplication/3'-3'";
	}

	// reference: get_fusion_strand (:616-635)
	std::string fusion_strand(bool strand, int gene, bool ambiguous) const {
		std::string result = genes[gene].is_dummy ? "." : (genes[gene].strand ? "+" : "-");
		result += "/";
		result += ambiguous ? "." : (strand ? "+" : "-");
		return result;
	}

	// reference: get_fusion_site (:637-709); the exons at the breakpoint are visited in the order of the reference's exon set (hazard H1)
	std::string fusion_site(int gene, bool spliced, bool exonic, contig_t contig, position_t breakpoint) const {
		const GeneRecord& record = genes[gene];
		if (record.is_dummy || breakpoint < record.start || breakpoint > record.end) return "intergenic";
		if (!exonic) return "intron";
		std::vector<uint32_t> exons;
		get_annotation_by_coordinate(contig, breakpoint, breakpoint, exons, exon_index);
		bool has_overlapping_exon = false, is_utr = true;
		unsigned is_3_end = 0, is_5_end = 0;
		for (size_t e = 0; e < exons.size(); ++e) {
			const ExonRecord& exon = annotation.exons[exons[e]];
			if (exon.gene != gene) continue;
			has_overlapping_exon = true;
			if (exon.coding_region_start <= breakpoint && exon.coding_region_end >= breakpoint) is_utr = false;
			if (!is_utr || !record.is_protein_coding) continue;
			// 5' or 3' UTR: which comes first when walking away from the breakpoint, a coding exon or the end of the transcript?
			if (exon.coding_region_start != -1 && exon.coding_region_start > breakpoint) { if (record.strand) ++is_5_end; else ++is_3_end; }
			else if (exon.coding_region_end != -1 && exon.coding_region_end < breakpoint) { if (!record.strand) ++is_5_end; else ++is_3_end; }
			else {
				int next_exon = exon.next_exon;
				while (next_exon != -1 && annotation.exons[next_exon].coding_region_start == -1) next_exon = annotation.exons[next_exon].next_exon;
				int previous_exon = exon.previous_exon;
				while (previous_exon != -1 && annotation.exons[previous_exon].coding_region_start == -1) previous_exon = annotation.exons[previous_exon].previous_exon;
				if (previous_exon != -1 || next_exon != -1) { // the transcript has a coding region
					if ((next_exon == -1) != !record.strand) ++is_3_end; else ++is_5_end;
				}
			}
		}
		std::string site;
		if (!has_overlapping_exon) site = "intron";
		else if (!record.is_protein_coding) site = "exon";
		else if (!is_utr) site = "CDS";
		else if (is_3_end > is_5_end) site = "3'UTR";
		else if (is_3_end < is_5_end) site = "5'UTR";
		else if (is_3_end + is_5_end == 0) site = "exon";
		else site = "UTR";
		if (spliced && site != "intron") site += "/splice-site";
		return site;
	}

	// reference: matches_blacklist_item (source/filter_blacklisted_ranges.cpp:136-218) for the items a tags file can hold (gene, position, range)
	bool matches_item(const agpu_range_item& item, const Fusion& f, int which_breakpoint, int max_mate_gap) const {
		const int gene = which_breakpoint == 1 ? f.gene1 : f.gene2;
		if (item.type == AGPU_RULE_GENE) return gene == (int) item.gene;
		if (item.type != AGPU_RULE_POSITION && item.type != AGPU_RULE_RANGE) return false;
		if ((which_breakpoint == 1 ? f.contig1 : f.contig2) != item.contig) return false;
		if (item.strand_defined && !f.strands_ambiguous && (which_breakpoint == 1 ? f.predicted_strand1 : f.predicted_strand2) != (item.strand != 0)) return false;
		if (item.type == AGPU_RULE_RANGE) { // the gene of the breakpoint overlaps the range by more than half (overlapping_fraction, :121-133)
			const position_t start1 = genes[gene].start, end1 = genes[gene].end, start2 = item.start, end2 = item.end;
			float fraction = 0;
			if (start1 >= start2 && end1 <= end2) fraction = 1;
			else if (start1 < start2 && end1 > end2) fraction = 1.0 * (end2 - start2) / (end1 - start1 + 1);
			else if (start1 >= start2 && start1 <= end2) fraction = 1.0 * (end2 - start1) / (end1 - start1 + 1);
			else if (end1 >= start2 && end1 <= end2) fraction = 1.0 * (end1 - start2) / (end1 - start1 + 1);
			return fraction > 0.5;
		}
		const position_t breakpoint = which_breakpoint == 1 ? f.breakpoint1 : f.breakpoint2;
		if (breakpoint == item.start) return true;
		if (f.split_reads1 + f.split_reads2 == 0) { // discordant mates near the position and pointing towards it
			const bool upstream = which_breakpoint == 1 ? f.upstream1 : f.upstream2;
			if ((!upstream && breakpoint <= item.start && breakpoint >= item.start - max_mate_gap) || (upstream && breakpoint >= item.start && breakpoint <= item.start + max_mate_gap)) return true;
		}
		return false;
	}
	// reference: annotate_tags (source/annotate_tags.cpp:46-84)
	std::string tags_of(const Fusion& f, const Tags& tags, int max_mate_gap) const {
		const contig_t contig_of[4] = { f.contig1, f.contig2, f.contig1, f.contig2 };
		const position_t start_of[4] = { f.breakpoint1, f.breakpoint2, genes[f.gene1].start, genes[f.gene2].start }, end_of[4] = { f.breakpoint1, f.breakpoint2, genes[f.gene1].end, genes[f.gene2].end };
		const int gene_5 = f.transcript_start_gene1 ? 1 : 2, gene_3 = 3 - gene_5;
		std::set<std::string> matching;
		for (int range = 0; range < 4; ++range)
			for (position_t bin = start_of[range] / 100000; bin <= (end_of[range] + 100000 - 1) / 100000; ++bin) {
				const std::map<uint64_t, std::vector<uint32_t> >::const_iterator candidates = tags.by_bin.find((uint64_t) contig_of[range] << 32 | (uint32_t) (bin * 100000));
				if (candidates == tags.by_bin.end()) continue;
				for (size_t k = 0; k < candidates->second.size(); ++k) {
					const TagRule& rule = tags.rules[candidates->second[k]];
					if (matches_item(rule.first, f, gene_5, max_mate_gap) && matches_item(rule.second, f, gene_3, max_mate_gap)) matching.insert(rule.tag);
				}
			}
		std::string result;
		for (std::set<std::string>::const_iterator tag = matching.begin(); tag != matching.end(); ++tag) { if (!result.empty()) result += ","; result += *tag; }
		return result.empty() ? "." : result;
	}
	// reference: annotate_retained_protein_domains (source/annotate_protein_domains.cpp:123-160): the domains of the gene on the retained side of the breakpoint;
	// a domain counts once per index bucket it spans, as in the reference (domains of one name add up)
	std::string retained_domains(contig_t contig, position_t breakpoint, bool predicted_strand, bool strand_ambiguous, int gene, bool upstream, const std::vector<ProteinDomain>& domains, const FlatIndex& index) const {
		const GeneRecord& record = genes[gene];
		if (!record.is_protein_coding || strand_ambiguous || predicted_strand != record.strand || (size_t) contig >= index.n_contigs()) return "";
		std::map<std::string, std::pair<unsigned, unsigned> > retained; // name -> (length, retained bases)
		for (uint32_t bucket = index.lower_bound(contig, record.start); bucket != index.contig_end(contig) && index.keys[bucket] <= record.end; ++bucket)
			for (uint32_t m = index.member_offset[bucket]; m < index.member_offset[bucket + 1]; ++m) {
				const ProteinDomain& domain = domains[index.members[m]];
				if (domain.gene != gene) continue;
				unsigned retained_bases = 0;
				if (upstream && domain.end >= breakpoint) retained_bases = domain.end - std::max(domain.start, breakpoint) + 1;
				else if (!upstream && domain.start <= breakpoint) retained_bases = std::min(domain.end, breakpoint) - domain.start + 1;
				std::pair<unsigned, unsigned>& entry = retained[domain.name];
				entry.first += domain.end - domain.start + 1; entry.second += retained_bases;
			}
		std::string result;
		for (std::map<std::string, std::pair<unsigned, unsigned> >::const_iterator domain = retained.begin(); domain != retained.end(); ++domain)
			if (domain->second.second > 0) { if (!result.empty()) result += ","; result += domain->first + "(" + std::to_string((long long) (domain->second.second * 100 / domain->second.first)) + "%)"; }
		return result;
	}

	Fusion row(uint32_t c) const {
		Fusion f;
		const uint32_t flags = table.flags[c];
		f.candidate = c; f.gene1 = (int) table.gene1[c]; f.gene2 = (int) table.gene2[c]; f.contig1 = (contig_t) (table.contigs[c] >> 16); f.contig2 = (contig_t) (table.contigs[c] & 0xFFFF);
		f.breakpoint1 = table.breakpoint1[c]; f.breakpoint2 = table.breakpoint2[c];
		f.upstream1 = flags & AGPU_CFLAG_UPSTREAM1; f.upstream2 = flags & AGPU_CFLAG_UPSTREAM2; f.exonic1 = flags & AGPU_CFLAG_EXONIC1; f.exonic2 = flags & AGPU_CFLAG_EXONIC2;
		f.spliced1 = flags & AGPU_CFLAG_SPLICED1; f.spliced2 = flags & AGPU_CFLAG_SPLICED2; f.predicted_strand1 = flags & AGPU_CFLAG_PREDICTED_STRAND1; f.predicted_strand2 = flags & AGPU_CFLAG_PREDICTED_STRAND2;
		f.strands_ambiguous = flags & AGPU_CFLAG_PREDICTED_STRANDS_AMBIGUOUS; f.transcript_start_gene1 = flags & AGPU_CFLAG_TRANSCRIPT_START_GENE1; f.transcript_start_ambiguous = flags & AGPU_CFLAG_TRANSCRIPT_START_AMBIGUOUS;
		f.split_reads1 = table.split_reads1[c]; f.split_reads2 = table.split_reads2[c]; f.discordant_mates = table.discordant_mates[c];
		f.filter = table.filter[c]; f.confidence = table.confidence[c]; f.evalue = table.evalue[c];
		return f;
	}
};

std::string coverage_text(int coverage) { return coverage >= 0 ? std::to_string((long long) coverage) : "."; }

}

void write_fusions_to_file(const Annotation& annotation, const FlatIndex& exon_index, const Contigs& contigs, const Assembly& assembly, const Coverage& coverage, const Batch* batch, const FusionTable& table,
                           const std::string& path, bool write_discarded, bool print_extra_info, unsigned max_itd_length, const OutputExtras& extras) {
	const std::chrono::steady_clock::time_point profile_start = std::chrono::steady_clock::now();
	const bool profile = getenv("ARRIBA_WRITER_PROFILE") != NULL;
	auto profile_mark = [&](const char* what) { if (profile) fprintf(stderr, "[writer] %s: %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - profile_start).count()); };
	Writer writer = { annotation, contigs, coverage, table, std::vector<GeneRecord>(), FlatIndex(), exon_index };
	// the gene records of this sample: the GTF genes, then the dummy genes the device cut from the unmapped positions
	if (table.n_genes < annotation.real_genes) throw std::runtime_error("gene table smaller than the annotation");
	writer.genes.assign(annotation.genes.begin(), annotation.genes.begin() + annotation.real_genes);
	for (uint32_t g = (uint32_t) annotation.real_genes; g < table.n_genes; ++g) {
		GeneRecord dummy;
		dummy.contig = table.gene_contig[g]; dummy.start = table.gene_start[g]; dummy.end = table.gene_end[g]; dummy.strand = true; dummy.exonic_length = 10000; dummy.is_dummy = true; dummy.is_protein_coding = false;
		writer.genes.push_back(dummy);
	}
	make_flat_index(writer.genes, std::max(writer.genes.size(), contigs.size()), writer.gene_index);

	std::vector<Fusion> rows;
	for (uint32_t c = 0; c < table.n_candidates; ++c)
		if (write_discarded != (table.filter[c] == 0)) rows.push_back(writer.row(c));
	if (write_discarded) {
		// not sorted: the iteration order of fusions_t
		std::sort(rows.begin(), rows.end(), [&](const Fusion& x, const Fusion& y) { return table.iteration_rank[x.candidate] < table.iteration_rank[y.candidate]; });
	} else {
		// events of one gene pair stay together, at the rank of the best of them (:1059-1074)
		std::map<std::pair<int, int>, const Fusion*> best_of_pair;
		for (size_t r = 0; r < rows.size(); ++r) {
			const Fusion*& best = best_of_pair[std::make_pair(rows[r].gene1, rows[r].gene2)];
			if (best == NULL || more_support(rows[r], *best)) best = &rows[r];
		}
		// (the best event of a row's gene pair is looked up once per row, not twice per comparison: 50 ms of a sort of 28 701 rows were map look-ups)
		std::vector<const Fusion*> best_of_row(rows.size());
		for (size_t r = 0; r < rows.size(); ++r) best_of_row[r] = best_of_pair.at(std::make_pair(rows[r].gene1, rows[r].gene2));
		std::vector<uint32_t> order(rows.size());
		for (size_t r = 0; r < rows.size(); ++r) order[r] = (uint32_t) r;
		std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
			const Fusion* best_x = best_of_row[x]; const Fusion* best_y = best_of_row[y];
			return best_x != best_y ? more_support(*best_x, *best_y) : more_support(rows[x], rows[y]);
		});
		std::vector<Fusion> sorted;
		sorted.reserve(rows.size());
		for (size_t r = 0; r < order.size(); ++r) sorted.push_back(rows[order[r]]);
		rows.swap(sorted);
	}

	profile_mark("genes indexed, rows sorted");
	const bool to_text = extras.text_of_part != NULL;
	if (to_text && extras.parts > 1) { // this rank's share of the rows, in their order
		std::vector<Fusion> mine;
		for (size_t r = extras.part; r < rows.size(); r += extras.parts) mine.push_back(rows[r]);
		rows.swap(mine);
	}
	// The file is written with pwrite where it can seek: the rows of a chunk go to their places from all threads at once (the 130 MB of a 10^8-fragment sample's fusions.tsv took
	// one thread 40 ms while the others waited); a pipe (-o /dev/stdout) gets them one after the other.
	const int out = to_text ? -1 : open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
	if (out < 0 && !to_text) throw std::runtime_error("failed to open output file");
	const bool can_seek = out >= 0 && lseek(out, 0, SEEK_CUR) != (off_t) -1;
	off_t file_offset = 0;
	auto write_all = [&](const char* data, size_t size, off_t at) -> bool { // (at: only where the file can seek)
		while (size > 0) {
			const ssize_t n = can_seek ? pwrite(out, data, size, at) : write(out, data, size);
			if (n < 0) { if (errno == EINTR) continue; return false; }
			data += n; size -= (size_t) n; at += n;
		}
		return true;
	};
	std::string text = "#gene1\tgene2\tstrand1(gene/fusion)\tstrand2(gene/fusion)\tbreakpoint1\tbreakpoint2\tsite1\tsite2\ttype\tsplit_reads1\tsplit_reads2\tdiscordant_mates\tcoverage1\tcoverage2\tconfidence\treading_frame\ttags\t"
	                   "retained_protein_domains\tclosest_genomic_breakpoint1\tclosest_genomic_breakpoint2\tgene_id1\tgene_id2\ttranscript_id1\ttranscript_id2\tdirection1\tdirection2\tfilters\tfusion_transcript\tpeptide_sequence\tread_identifiers\n";
	static const char* const confidence_names[] = { "low", "medium", "high", "high" };
	std::atomic<long long> profile_ns[3]; profile_ns[0] = 0; profile_ns[1] = 0; profile_ns[2] = 0; // (ARRIBA_WRITER_PROFILE: transcript from the pileups, best-fitting transcripts, peptides)
	auto format_row = [&](size_t r, std::string& text) {
		const Fusion& f = rows[r];
		std::string site_5 = writer.fusion_site(f.gene1, f.spliced1, f.exonic1, f.contig1, f.breakpoint1), site_3 = writer.fusion_site(f.gene2, f.spliced2, f.exonic2, f.contig2, f.breakpoint2);
		// the 5' gene comes first
		int gene_5 = f.gene1, gene_3 = f.gene2; contig_t contig_5 = f.contig1, contig_3 = f.contig2; position_t breakpoint_5 = f.breakpoint1, breakpoint_3 = f.breakpoint2;
		bool upstream_5 = f.upstream1, upstream_3 = f.upstream2, strand_5 = f.predicted_strand1, strand_3 = f.predicted_strand2; unsigned split_reads_5 = f.split_reads1, split_reads_3 = f.split_reads2;
		if (!f.transcript_start_gene1) {
			std::swap(gene_5, gene_3); std::swap(contig_5, contig_3); std::swap(breakpoint_5, breakpoint_3); std::swap(upstream_5, upstream_3); std::swap(strand_5, strand_3);
			std::swap(split_reads_5, split_reads_3); std::swap(site_5, site_3);
		}
		const int coverage_5 = coverage.get_coverage(contig_5, breakpoint_5, !upstream_5), coverage_3 = coverage.get_coverage(contig_3, breakpoint_3, !upstream_3);
		std::string transcript_sequence = ".", peptide_sequence = ".", reading_frame = ".", transcript_id_5 = ".", transcript_id_3 = ".";
		if (print_extra_info && batch != NULL) {
			// the fusion transcript from the pileups of the supporting reads, then the pair of annotated transcripts that gives an in-frame peptide (:1118-1154)
			const TranscriptInput input = { *batch, table.read_filter, assembly, annotation, exon_index };
			const uint64_t* offsets = table.list_offset + 3 * (size_t) f.candidate;
			FusionEvent event;
			event.contig_of_gene1 = writer.genes[f.gene1].contig; event.contig_of_gene2 = writer.genes[f.gene2].contig; event.breakpoint1 = f.breakpoint1; event.breakpoint2 = f.breakpoint2;
			event.upstream1 = f.upstream1; event.upstream2 = f.upstream2; event.predicted_strand1 = f.predicted_strand1; event.predicted_strand2 = f.predicted_strand2; event.strands_ambiguous = f.strands_ambiguous;
			event.transcript_start_gene1 = f.transcript_start_gene1; event.transcript_start_ambiguous = f.transcript_start_ambiguous;
			event.split_read1_list = table.read_lists + offsets[0]; event.split_read2_list = table.read_lists + offsets[1]; event.discordant_mate_list = table.read_lists + offsets[2];
			event.n_split_reads1 = (uint32_t) (offsets[1] - offsets[0]); event.n_split_reads2 = (uint32_t) (offsets[2] - offsets[1]); event.n_discordant_mates = (uint32_t) (offsets[3] - offsets[2]);
			std::vector<position_t> positions;
			const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
			fusion_transcript_sequence(input, event, transcript_sequence, positions);
			const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();
			if (profile) profile_ns[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
			std::vector<int> transcripts_5, transcripts_3;
			best_fitting_transcripts(input, transcript_sequence, positions, gene_5, writer.genes[gene_5].is_dummy, writer.genes[gene_5].contig, writer.genes[gene_5].strand, strand_5, f.strands_ambiguous, 5, transcripts_5);
			best_fitting_transcripts(input, transcript_sequence, positions, gene_3, writer.genes[gene_3].is_dummy, writer.genes[gene_3].contig, writer.genes[gene_3].strand, strand_3, f.strands_ambiguous, 3, transcripts_3);
			const PeptideGenes peptide_genes = { writer.genes[gene_5].contig, writer.genes[gene_3].contig, writer.genes[gene_5].strand, writer.genes[gene_3].strand, writer.genes[gene_5].is_dummy, writer.genes[gene_3].is_dummy, strand_3 };
			const std::chrono::steady_clock::time_point t2 = std::chrono::steady_clock::now();
			if (profile) profile_ns[1] += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
			int transcript_5 = -1, transcript_3 = -1;
			const std::string sequence_as_assembled = transcript_sequence; const std::vector<position_t> positions_as_assembled = positions;
			const bool is_itd = f.gene1 == f.gene2 && (unsigned) f.breakpoint2 - (unsigned) f.breakpoint1 < max_itd_length && f.upstream1 && !f.upstream2; // source/common.hpp:270-274
			// every combination until one is in-frame; without candidates on one side the loop body still runs once with no transcript on that side
			for (size_t i5 = 0; (transcripts_5.empty() || i5 != transcripts_5.size()) && reading_frame != "in-frame"; ++i5) {
				if (i5 != transcripts_5.size()) transcript_5 = transcripts_5[i5];
				for (size_t i3 = 0; (transcripts_3.empty() || i3 != transcripts_3.size()) && reading_frame != "in-frame"; ++i3) {
					if (i3 != transcripts_3.size()) transcript_3 = transcripts_3[i3];
					if (extras.fill_sequence_gaps) { // -I: for every pair of transcripts anew, from the sequence as assembled
						transcript_sequence = sequence_as_assembled; positions = positions_as_assembled;
						fill_gaps_in_fusion_transcript(input, transcript_sequence, positions, transcript_5, transcript_3, strand_5, strand_3, is_itd);
					}
					peptide_sequence = fusion_peptide_sequence(input, transcript_sequence, positions, peptide_genes, transcript_5, transcript_3);
					reading_frame = reading_frame_verdict(peptide_sequence);
					if (i3 == transcripts_3.size()) break;
				}
				if (i5 == transcripts_5.size() || transcripts_3.empty()) break;
			}
			if (profile) profile_ns[2] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t2).count();
			if (reading_frame == "stop-codon") peptide_sequence = "."; // a stop codon in front of the junction: no peptide
			if (transcript_5 != -1) transcript_id_5 = annotation.transcripts[transcript_5].name;
			if (transcript_3 != -1) transcript_id_3 = annotation.transcripts[transcript_3].name;
		}
		text += writer.gene_to_name(gene_5, contig_5, breakpoint_5) + "\t" + writer.gene_to_name(gene_3, contig_3, breakpoint_3) + "\t";
		text += writer.fusion_strand(strand_5, gene_5, f.strands_ambiguous) + "\t" + writer.fusion_strand(strand_3, gene_3, f.strands_ambiguous) + "\t";
		text += contigs.original_names[contig_5] + ":" + std::to_string(breakpoint_5 + 1) + "\t" + contigs.original_names[contig_3] + ":" + std::to_string(breakpoint_3 + 1) + "\t";
		text += site_5 + "\t" + site_3 + "\t" + writer.fusion_type(f, max_itd_length) + "\t" + std::to_string(split_reads_5) + "\t" + std::to_string(split_reads_3) + "\t" + std::to_string(f.discordant_mates) + "\t";
		text += coverage_text(coverage_5) + "\t" + coverage_text(coverage_3) + "\t" + confidence_names[f.confidence & 3] + "\t" + reading_frame;
		text += "\t" + (extras.tags != NULL && !extras.tags->empty() ? writer.tags_of(f, *extras.tags, extras.max_mate_gap) : std::string("."));
		std::string domains = ".";
		if (extras.protein_domains != NULL && !extras.protein_domains->empty()) {
			const std::string domains_5 = writer.retained_domains(contig_5, breakpoint_5, strand_5, f.strands_ambiguous, gene_5, upstream_5, *extras.protein_domains, *extras.protein_domain_index);
			const std::string domains_3 = writer.retained_domains(contig_3, breakpoint_3, strand_3, f.strands_ambiguous, gene_3, upstream_3, *extras.protein_domains, *extras.protein_domain_index);
			if (!domains_5.empty() || !domains_3.empty()) domains = domains_5 + "|" + domains_3;
		}
		// closest genomic breakpoints as <contig>:<position>(<distance to the transcriptomic breakpoint>) (:1193-1203)
		position_t genomic_5 = table.closest_genomic_breakpoint1 != NULL ? table.closest_genomic_breakpoint1[f.candidate] : -1, genomic_3 = table.closest_genomic_breakpoint2 != NULL ? table.closest_genomic_breakpoint2[f.candidate] : -1;
		if (!f.transcript_start_gene1) std::swap(genomic_5, genomic_3);
		text += "\t" + domains;
		text += "\t" + (genomic_5 >= 0 ? contigs.original_names[contig_5] + ":" + std::to_string((long long) genomic_5 + 1) + "(" + std::to_string((long long) abs(breakpoint_5 - genomic_5)) + ")" : std::string("."));
		text += "\t" + (genomic_3 >= 0 ? contigs.original_names[contig_3] + ":" + std::to_string((long long) genomic_3 + 1) + "(" + std::to_string((long long) abs(breakpoint_3 - genomic_3)) + ")" : std::string("."));
		// reads discarded by a filter, by name of the filter
		std::map<std::string, unsigned> filters;
		if (f.filter != 0) filters[f.filter < N_FILTER_NAMES ? FILTER_NAMES[f.filter] : "?"] = 0;
		const uint64_t list_begin = table.list_offset[3 * (size_t) f.candidate], list_end = table.list_offset[3 * (size_t) f.candidate + 3];
		for (uint64_t k = list_begin; k < list_end; ++k) {
			const uint8_t read_filter = table.read_filter[table.read_lists[k]];
			if (read_filter != 0) filters[read_filter < N_FILTER_NAMES ? FILTER_NAMES[read_filter] : "?"]++;
		}
		text += "\t" + (writer.genes[gene_5].is_dummy ? std::string(".") : writer.genes[gene_5].gene_id) + "\t" + (writer.genes[gene_3].is_dummy ? std::string(".") : writer.genes[gene_3].gene_id);
		text += "\t" + transcript_id_5 + "\t" + transcript_id_3;
		text += std::string("\t") + (upstream_5 ? "upstream" : "downstream") + "\t" + (upstream_3 ? "upstream" : "downstream") + "\t";
		if (filters.empty()) text += ".";
		else
			for (std::map<std::string, unsigned>::const_iterator filter = filters.begin(); filter != filters.end(); ++filter) {
				if (filter != filters.begin()) text += ",";
				text += filter->first;
				if (filter->second != 0) text += "(" + std::to_string(filter->second) + ")";
			}
		text += "\t" + transcript_sequence + "\t" + peptide_sequence + "\t";
		if (print_extra_info && list_end > list_begin && batch != NULL) {
			// "QNAME,HI": the HI tag goes (name.substr(0, name.find_last_of(','))), straight from the pool of names -- a fusion of a deeply sequenced sample lists 10^5 reads,
			// the file of a 10^8-fragment sample 9.3 M, and two temporary strings per read were a third of the writer's thread time
			const char* names = batch->names.data();
			size_t bytes = 0;
			for (uint64_t k = list_begin; k < list_end; ++k) { const uint32_t read = table.read_lists[k]; bytes += batch->name_offset[read + 1] - batch->name_offset[read] + 1; }
			text.reserve(text.size() + bytes + 1);
			for (uint64_t k = list_begin; k < list_end; ++k) {
				if (k != list_begin) text += ',';
				const uint32_t read = table.read_lists[k];
				const char* begin = names + batch->name_offset[read]; const char* end = names + batch->name_offset[read + 1];
				const char* comma = end;
				while (comma > begin && comma[-1] != ',') --comma;
				text.append(begin, comma > begin ? (size_t) (comma - 1 - begin) : (size_t) (end - begin));
			}
		} else text += ".";
		text += "\n";
	};
	// the rows are independent of each other: all threads format a chunk of them (the fusion transcripts from the read pileups are the expensive part), the
	// chunk is written in order, the warnings of a row come out in row order as well
	const size_t CHUNK = 16384;
	unsigned int n_threads = cpu_budget();
	if (const char* setting = getenv("ARRIBA_WRITER_THREADS")) if (atoi(setting) > 0) n_threads = (unsigned int) atoi(setting);
	n_threads = std::max(1u, std::min(n_threads, 128u));
	std::vector<std::string> row_text, row_warnings;
	for (size_t chunk_begin = 0; chunk_begin < rows.size(); chunk_begin += CHUNK) {
		const size_t chunk_end = std::min(rows.size(), chunk_begin + CHUNK), count = chunk_end - chunk_begin;
		row_text.assign(count, std::string()); row_warnings.assign(count, std::string());
		std::atomic<size_t> next(0);
		std::exception_ptr failure;
		std::mutex failure_mutex;
		auto work = [&] {
			for (size_t k = next.fetch_add(1); k < count; k = next.fetch_add(1)) {
				transcript_warnings = &row_warnings[k];
				const std::chrono::steady_clock::time_point row_start = std::chrono::steady_clock::now();
				try { format_row(chunk_begin + k, row_text[k]); }
				catch (...) { std::lock_guard<std::mutex> lock(failure_mutex); if (!failure) failure = std::current_exception(); }
				transcript_warnings = NULL;
				if (profile) { const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - row_start).count(); if (seconds > 0.02) fprintf(stderr, "[writer] row %zu took %.3f s (%u + %u + %u supporting reads)\n", chunk_begin + k, seconds, rows[chunk_begin + k].split_reads1, rows[chunk_begin + k].split_reads2, rows[chunk_begin + k].discordant_mates); }
			}
		};
		if (n_threads == 1 || count < 4) work();
		else { std::lock_guard<std::mutex> one_file(formatter_pool_in_use); formatter_pool().run((unsigned) std::min<size_t>(n_threads, count), work); }
		if (failure) { if (out >= 0) close(out); std::rethrow_exception(failure); }
		for (size_t k = 0; k < count; ++k) {
			if (!row_warnings[k].empty()) fputs(row_warnings[k].c_str(), stderr);
			if (to_text) text += row_text[k]; // (collected below)
		}
		if (to_text) continue;
		// (the rows go to the file as they are: joined into one string first, the 130 MB of a 10^8-fragment sample's file were copied once more)
		bool written = write_all(text.data(), text.size(), file_offset); // (the header line, in front of the first chunk)
		file_offset += (off_t) text.size();
		text.clear();
		if (can_seek && n_threads > 1 && count >= 4) {
			std::vector<off_t> row_offset(count);
			for (size_t k = 0; k < count; ++k) { row_offset[k] = file_offset; file_offset += (off_t) row_text[k].size(); }
			std::atomic<size_t> next_row(0);
			std::atomic<bool> all_written(written);
			auto write_rows = [&] { for (size_t k = next_row.fetch_add(64); k < count; k = next_row.fetch_add(64)) for (size_t r = k; r < std::min(count, k + 64); ++r) if (!write_all(row_text[r].data(), row_text[r].size(), row_offset[r])) all_written = false; };
			{ std::lock_guard<std::mutex> one_file(formatter_pool_in_use); formatter_pool().run((unsigned) std::min<size_t>(n_threads, (count + 63) / 64), write_rows); }
			written = all_written;
		} else
			for (size_t k = 0; k < count && written; ++k) { written = write_all(row_text[k].data(), row_text[k].size(), file_offset); file_offset += (off_t) row_text[k].size(); }
		if (!written) { close(out); throw std::runtime_error("failed to write to file"); }
	}
	profile_mark("rows formatted and written");
	if (to_text) {
		if (extras.part != 0) text.erase(0, text.find('\n') + 1); // the header line travels with part 0
		extras.text_of_part->append(text);
		return;
	}
	if (profile) {
		fprintf(stderr, "[writer] thread time: fusion transcripts %.3f s, best-fitting transcripts %.3f s, peptides %.3f s\n", profile_ns[0] * 1e-9, profile_ns[1] * 1e-9, profile_ns[2] * 1e-9);
		fprintf(stderr, "[writer] fusion transcripts: pile-ups %.3f s, consensus %.3f s (of it the columns of the pile-ups %.3f s), the rest %.3f s\n", transcript_profile_ns[0] * 1e-9, transcript_profile_ns[2] * 1e-9, transcript_profile_ns[1] * 1e-9, transcript_profile_ns[3] * 1e-9);
		for (int k = 0; k < 4; ++k) transcript_profile_ns[k] = 0;
	}
	const bool ok = write_all(text.data(), text.size(), file_offset); // (a file without rows: the header line)
	if (close(out) != 0 || !ok) throw std::runtime_error("failed to write to file");
}

}
