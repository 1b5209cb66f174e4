// tools/gen_synth.cpp -- see gen_synth.hpp.  Test + bench tooling, not product code.
#include "gen_synth.hpp"
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <zlib.h>

namespace synth {

struct Rng {
	uint64_t state;
	explicit Rng(uint64_t seed): state(seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL) {}
	uint64_t next() { // splitmix64
		uint64_t z = (state += 0x9E3779B97F4A7C15ULL);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
		return z ^ (z >> 31);
	}
	uint32_t below(uint32_t n) { return (uint32_t) (((next() >> 32) * (uint64_t) n) >> 32); }
	int range(int lo, int hi) { return lo + (int) below((uint32_t) (hi - lo + 1)); } // inclusive
	double unif() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
	bool chance(double p) { return unif() < p; }
};

namespace {

const uint16_t F_PAIRED = 1, F_PROPER = 2, F_REVERSE = 16, F_MREVERSE = 32, F_READ1 = 64, F_READ2 = 128, F_SECONDARY = 256, F_DUP = 1024, F_SUPPLEMENTARY = 2048;
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5 };
inline uint32_t cig(uint32_t length, uint32_t op) { return length << 4 | op; }

struct Side { int contig; int bp; bool upstream; int gene; }; // gene = -1: no transcript context
struct Junction { Side a, b; };
struct Aln { int contig; int start, end; std::vector<uint32_t> cigar; std::string seq; };
struct Record {
	int hi, nh;
	uint16_t flag;
	int contig, pos;
	std::vector<uint32_t> cigar;
	std::string seq;
	bool sa;
	bool no_hi = false;
};
typedef std::vector<Record> Fragment;

char complement(char c) {
	switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; }
}
std::string revcomp(const std::string& s) {
	std::string r(s.size(), 'N');
	for (size_t i = 0; i < s.size(); ++i)
		r[s.size() - 1 - i] = complement(s[i]);
	return r;
}

}

struct Generator::Impl {
	Rng rng;
	std::vector<Junction> junctions;
	std::vector<double> junction_cdf;
	std::vector<double> gene_cdf;
	std::vector<std::pair<int,int> > read_through_pairs; // (left gene, right gene) on the same strand, close together
	std::vector<std::vector<int> > genes_by_contig;
	struct ItdHotspot { int gene, q, d; };
	std::vector<ItdHotspot> itd_hotspots;
	int n_main_contigs = 0, viral_contig = -1, viral_contig2 = -1, boring_contig = -1;
	explicit Impl(uint64_t seed): rng(seed) {}
};

Generator::Generator(const Config& config): config_(config), impl_(new Impl(config.seed)) {}

// ---------------------------------------------------------------------------------------------
// reference: genome, genes, junctions
// ---------------------------------------------------------------------------------------------

static char random_base(Rng& rng) {
	double u = rng.unif();
	return (u < 0.295) ? 'A' : (u < 0.59) ? 'T' : (u < 0.795) ? 'C' : 'G';
}

void Generator::build_reference() {
	Rng& rng = impl_->rng;
	const Config& c = config_;

	// contigs
	impl_->n_main_contigs = c.contigs;
	for (int i = 0; i < c.contigs; ++i) {
		contig_names_.push_back((i < 22) ? std::to_string(i + 1) : (i == 22 ? "X" : (i == 23 ? "Y" : "U" + std::to_string(i))));
		int length = (int) (c.contig_length * (0.8 + 0.4 * rng.unif()));
		std::string sequence(length, 'A');
		for (int p = 0; p < length; ++p)
			sequence[p] = random_base(rng);
		contig_sequences_.push_back(sequence);
	}
	if (c.viral) {
		impl_->viral_contig = contig_names_.size();
		contig_names_.push_back("NC_001526.4");
		std::string v(7900, 'A');
		for (size_t p = 0; p < v.size(); ++p) v[p] = random_base(rng);
		contig_sequences_.push_back(v);
		impl_->viral_contig2 = contig_names_.size();
		contig_names_.push_back("AC_000007.1");
		std::string v2(12000, 'A');
		for (size_t p = 0; p < v2.size(); ++p) v2[p] = random_base(rng);
		contig_sequences_.push_back(v2);
	}
	impl_->boring_contig = contig_names_.size();
	contig_names_.push_back("GL000220.1");
	{
		std::string b(60000, 'A');
		for (size_t p = 0; p < b.size(); ++p) b[p] = random_base(rng);
		contig_sequences_.push_back(b);
	}

	// genes
	impl_->genes_by_contig.resize(contig_names_.size());
	int gene_serial = 0, transcript_serial = 0;
	for (int contig = 0; contig < c.contigs; ++contig) {
		int length = contig_sequences_[contig].size();
		double mean_spacing = 1e6 / c.genes_per_mb;
		int cursor = 20000;
		while (cursor < length - 60000) {
			Gene gene;
			gene.contig = contig;
			gene.plus = rng.chance(0.5);
			int n_exons = rng.chance(0.05) ? 1 : rng.range(2, 10);
			Transcript full;
			int p = cursor;
			for (int e = 0; e < n_exons; ++e) {
				Exon exon;
				exon.start = p;
				exon.end = p + rng.range(60, 500) - 1;
				full.exons.push_back(exon);
				p = exon.end + 1 + rng.range(150, rng.chance(0.2) ? 6000 : 1500);
			}
			if (full.exons.back().end >= length - 30000)
				break;
			gene.start = full.exons.front().start;
			gene.end = full.exons.back().end;
			char buffer[64];
			snprintf(buffer, sizeof(buffer), "ENSG%011d.%d", ++gene_serial, rng.range(1, 15));
			gene.id = buffer;
			gene.name = "SYN" + std::to_string(gene_serial);
			bool coding = rng.chance(0.8);
			if (coding) {
				const Exon& first = full.exons[(n_exons > 2 && rng.chance(0.3)) ? 1 : 0];
				const Exon& last = full.exons[(n_exons > 3 && rng.chance(0.3)) ? n_exons - 2 : n_exons - 1];
				full.cds_start = rng.range(first.start, first.start + (first.end - first.start) / 2);
				full.cds_end = rng.range(last.start + (last.end - last.start) / 2, last.end);
				if (rng.chance(0.1)) full.cds_start = first.start; // incomplete annotation: first base of exon is coding
				if (rng.chance(0.1)) full.cds_end = last.end;
			}
			snprintf(buffer, sizeof(buffer), "ENST%011d.%d", ++transcript_serial, rng.range(1, 9));
			full.id = buffer;
			gene.transcripts.push_back(full);
			// alternative transcripts: skip internal exons
			int n_alternative = (n_exons >= 4) ? rng.range(0, 2) : 0;
			for (int a = 0; a < n_alternative; ++a) {
				Transcript alternative;
				for (int e = 0; e < n_exons; ++e)
					if (e == 0 || e == n_exons - 1 || !rng.chance(0.35))
						alternative.exons.push_back(full.exons[e]);
				if (alternative.exons.size() == full.exons.size())
					continue;
				alternative.cds_start = full.cds_start;
				alternative.cds_end = full.cds_end;
				snprintf(buffer, sizeof(buffer), "ENST%011d.%d", ++transcript_serial, rng.range(1, 9));
				alternative.id = buffer;
				gene.transcripts.push_back(alternative);
			}
			impl_->genes_by_contig[contig].push_back(genes_.size());
			genes_.push_back(gene);

			// overlapping antisense gene
			if (rng.chance(0.10) && gene.end - gene.start > 2000) {
				Gene antisense;
				antisense.contig = contig;
				antisense.plus = !gene.plus;
				Transcript t;
				int q = rng.range(gene.start + 100, gene.start + (gene.end - gene.start) / 2);
				int m = rng.range(1, 4);
				for (int e = 0; e < m; ++e) {
					Exon exon;
					exon.start = q;
					exon.end = q + rng.range(80, 400) - 1;
					t.exons.push_back(exon);
					q = exon.end + 1 + rng.range(150, 1200);
				}
				antisense.start = t.exons.front().start;
				antisense.end = t.exons.back().end;
				if (antisense.end < length - 30000) {
					snprintf(buffer, sizeof(buffer), "ENSG%011d.%d", ++gene_serial, rng.range(1, 15));
					antisense.id = buffer;
					antisense.name = "SYN" + std::to_string(gene_serial) + "-AS1";
					snprintf(buffer, sizeof(buffer), "ENST%011d.%d", ++transcript_serial, rng.range(1, 9));
					t.id = buffer;
					antisense.transcripts.push_back(t);
					impl_->genes_by_contig[contig].push_back(genes_.size());
					genes_.push_back(antisense);
					p = std::max(p, antisense.end + 1);
				}
			}
			// a pile of overlapping genes on the same locus (gene families, read-through transcripts): gene sets of up to gene_stack + 1 ids.
			// (no random numbers are drawn when the option is off: the default datasets stay byte-identical)
			if (c.gene_stack > 0 && rng.chance(0.2)) {
				const int copies = rng.range(2, c.gene_stack);
				for (int copy = 1; copy <= copies; ++copy) {
					Gene stacked;
					stacked.contig = contig;
					stacked.plus = rng.chance(0.7) ? gene.plus : !gene.plus;
					Transcript t;
					const int shift = rng.range(3, 40) * copy;
					for (size_t e = 0; e < full.exons.size(); ++e) {
						if (e > 0 && e + 1 < full.exons.size() && rng.chance(0.2)) continue;
						Exon exon;
						exon.start = full.exons[e].start + shift;
						exon.end = full.exons[e].end + shift;
						t.exons.push_back(exon);
					}
					stacked.start = t.exons.front().start;
					stacked.end = t.exons.back().end;
					if (stacked.end >= length - 30000) break;
					snprintf(buffer, sizeof(buffer), "ENSG%011d.%d", ++gene_serial, rng.range(1, 15));
					stacked.id = buffer;
					stacked.name = "SYN" + std::to_string(gene_serial) + "-L" + std::to_string(copy);
					snprintf(buffer, sizeof(buffer), "ENST%011d.%d", ++transcript_serial, rng.range(1, 9));
					t.id = buffer;
					stacked.transcripts.push_back(t);
					impl_->genes_by_contig[contig].push_back(genes_.size());
					genes_.push_back(stacked);
					p = std::max(p, stacked.end + 1);
				}
			}
			cursor = std::max(p, gene.end) + (int) (mean_spacing * (0.1 + 1.5 * rng.unif()));
		}
	}
	if (genes_.size() < 4)
		throw std::runtime_error("too few genes; increase contig_length or genes_per_mb");

	// N blocks in intergenic space (about 2% of the genome)
	for (int contig = 0; contig < c.contigs; ++contig) {
		std::string& sequence = contig_sequences_[contig];
		long target = sequence.size() / 50, done = 0;
		int attempts = 0;
		while (done < target && attempts++ < 1000) {
			int block = rng.range(500, 5000);
			int start = rng.range(1000, (int) sequence.size() - block - 1000);
			bool overlaps_gene = false;
			for (size_t g = 0; g < impl_->genes_by_contig[contig].size() && !overlaps_gene; ++g) {
				const Gene& gene = genes_[impl_->genes_by_contig[contig][g]];
				overlaps_gene = start <= gene.end + 3000 && start + block >= gene.start - 3000;
			}
			if (overlaps_gene)
				continue;
			for (int p = start; p < start + block; ++p)
				sequence[p] = 'N';
			done += block;
		}
	}

	// paralogous genes: exon sequence copied (with ~3% edits) from another gene
	for (size_t g = 0; g < genes_.size(); ++g) {
		if (!rng.chance(c.frac_paralog_genes))
			continue;
		const Gene& source = genes_[rng.below(genes_.size())];
		if (&source == &genes_[g])
			continue;
		const Transcript& from = source.transcripts[0];
		const Transcript& to = genes_[g].transcripts[0];
		for (size_t e = 0; e < to.exons.size(); ++e) {
			const Exon& source_exon = from.exons[e % from.exons.size()];
			int n = std::min(to.exons[e].end - to.exons[e].start + 1, source_exon.end - source_exon.start + 1);
			for (int i = 0; i < n; ++i) {
				char base = contig_sequences_[source.contig][source_exon.start + i];
				if (rng.chance(0.03))
					base = random_base(rng);
				contig_sequences_[genes_[g].contig][to.exons[e].start + i] = base;
			}
		}
	}

	// gene expression weights (Zipf) for ordinary read pairs
	{
		std::vector<int> order(genes_.size());
		for (size_t i = 0; i < order.size(); ++i) order[i] = i;
		for (size_t i = order.size() - 1; i > 0; --i) std::swap(order[i], order[rng.below(i + 1)]);
		std::vector<double> weight(genes_.size());
		for (size_t rank = 0; rank < order.size(); ++rank)
			weight[order[rank]] = 1.0 / std::pow(rank + 1.0, 0.9);
		double total = 0;
		impl_->gene_cdf.resize(genes_.size());
		for (size_t i = 0; i < weight.size(); ++i) { total += weight[i]; impl_->gene_cdf[i] = total; }
		for (size_t i = 0; i < weight.size(); ++i) impl_->gene_cdf[i] /= total;
	}

	// recurrent internal tandem duplications: a fixed segment [q, q + d) inside a coding exon (drawn from a generator of their own)
	if (c.itd_hotspots > 0) {
		Rng hotspot_rng(c.seed ^ 0x17D5ULL);
		for (int attempt = 0; attempt < 100000 && (int) impl_->itd_hotspots.size() < c.itd_hotspots; ++attempt) {
			const int g = (int) hotspot_rng.below(genes_.size());
			const Transcript& t = genes_[g].transcripts[0];
			if (t.cds_start < 0) continue;
			const Exon& exon = t.exons[hotspot_rng.below(t.exons.size())];
			if (exon.end - exon.start < 260 || exon.start < t.cds_start || exon.end > t.cds_end) continue;
			Impl::ItdHotspot hotspot;
			hotspot.gene = g; hotspot.d = hotspot_rng.range(18, 48); hotspot.q = hotspot_rng.range(exon.start + 90, exon.end - hotspot.d - 90);
			impl_->itd_hotspots.push_back(hotspot);
		}
	}

	// neighbouring genes on the same strand (read-through candidates)
	for (int contig = 0; contig < c.contigs; ++contig) {
		const std::vector<int>& list = impl_->genes_by_contig[contig];
		for (size_t i = 0; i + 1 < list.size(); ++i) {
			const Gene& left = genes_[list[i]];
			const Gene& right = genes_[list[i + 1]];
			if (left.plus == right.plus && right.start > left.end && right.start - left.end < 40000 &&
			    left.transcripts[0].exons.size() >= 2 && right.transcripts[0].exons.size() >= 2)
				impl_->read_through_pairs.push_back(std::make_pair(list[i], list[i + 1]));
		}
	}

	// recurrent junctions
	auto make_junction = [&](Rng& rng, int gene_a, int gene_b) {
		Junction junction;
		bool spliced = rng.chance(0.75);
		for (int which = 0; which < 2; ++which) {
			int g = which == 0 ? gene_a : gene_b;
			const Gene& gene = genes_[g];
			const Transcript& t = gene.transcripts[0];
			Side side;
			side.contig = gene.contig;
			side.gene = g;
			bool five_prime = which == 0;
			if (spliced && t.exons.size() >= 2) {
				if (five_prime) { // donor: end of a non-terminal exon in transcription direction
					if (gene.plus) { int e = rng.range(0, t.exons.size() - 2); side.bp = t.exons[e].end; side.upstream = false; }
					else { int e = rng.range(1, t.exons.size() - 1); side.bp = t.exons[e].start; side.upstream = true; }
				} else { // acceptor: start of a non-first exon in transcription direction
					if (gene.plus) { int e = rng.range(1, t.exons.size() - 1); side.bp = t.exons[e].start; side.upstream = true; }
					else { int e = rng.range(0, t.exons.size() - 2); side.bp = t.exons[e].end; side.upstream = false; }
				}
			} else {
				side.bp = rng.range(gene.start + 30, gene.end - 30);
				side.upstream = rng.chance(0.5);
			}
			(which == 0 ? junction.a : junction.b) = side;
		}
		// low-complexity sequence right at the breakpoint (only meaningful for non-spliced junctions)
		if (rng.chance(c.frac_low_complexity_junctions * (spliced ? 0.3 : 3.0))) {
			Side& side = rng.chance(0.5) ? junction.a : junction.b;
			std::string& sequence = contig_sequences_[side.contig];
			int n = rng.range(8, 45);
			int kind = rng.range(0, 2);
			for (int i = 0; i < n; ++i) {
				int p = side.upstream ? side.bp + i : side.bp - i;
				if (p < 0 || p >= (int) sequence.size()) break;
				sequence[p] = (kind == 0) ? 'A' : (kind == 1) ? "CA"[i % 2] : "CAG"[i % 3];
			}
		}
		return junction;
	};
	for (int j = 0; j < c.junctions; ++j) {
		int gene_a = rng.below(genes_.size());
		int gene_b = rng.below(genes_.size());
		for (int attempt = 0; attempt < 20 && (gene_b == gene_a || (rng.chance(0.7) && genes_[gene_b].contig == genes_[gene_a].contig)); ++attempt)
			gene_b = rng.below(genes_.size());
		if (gene_b == gene_a)
			gene_b = (gene_a + 1) % genes_.size();
		impl_->junctions.push_back(make_junction(rng, gene_a, gene_b));
	}

	// families of homologous genes: the whole locus of one gene copied (with ~2 % edits, reverse-complemented between genes on different strands) over
	// the locus of another, plus well supported junctions gene-partner, homolog-partner and gene-homolog (drawn from a generator of their own)
	if (c.homolog_families > 0) {
		Rng family_rng(c.seed ^ 0x40F0106ULL);
		int made = 0;
		for (int attempt = 0; attempt < 100000 && made < c.homolog_families; ++attempt) {
			const int a = (int) family_rng.below(genes_.size()), b = (int) family_rng.below(genes_.size()), partner = (int) family_rng.below(genes_.size());
			if (a == b || a == partner || b == partner) continue;
			const Gene& from = genes_[a];
			const Gene& to = genes_[b];
			if (from.contig == to.contig && from.start <= to.end + 1000 && to.start <= from.end + 1000) continue;
			bool clear = true; // neither locus may carry another gene (its sequence would change under it)
			for (size_t g = 0; g < genes_.size() && clear; ++g)
				if ((int) g != b && genes_[g].contig == to.contig && genes_[g].start <= to.end && to.start <= genes_[g].end) clear = false;
			if (!clear) continue;
			const int n = std::min(from.end - from.start, to.end - to.start) + 1;
			const bool reverse = from.plus != to.plus;
			for (int i = 0; i < n; ++i) {
				char base = reverse ? complement(contig_sequences_[from.contig][from.start + n - 1 - i]) : contig_sequences_[from.contig][from.start + i];
				if (family_rng.chance(0.02)) base = random_base(family_rng);
				contig_sequences_[to.contig][to.start + i] = base;
			}
			const Junction family[3] = { make_junction(family_rng, a, partner), make_junction(family_rng, b, partner), make_junction(family_rng, a, b) };
			for (int k = 0; k < 3; ++k) {
				const size_t at = std::min<size_t>(impl_->junctions.size(), (size_t) (2 + 4 * made + k + (k == 1 ? 6 : 0))); // near the head of the Zipf ranking, unequal support
				impl_->junctions.insert(impl_->junctions.begin() + at, family[k]);
			}
			++made;
		}
	}
	{
		double total = 0;
		impl_->junction_cdf.resize(impl_->junctions.size());
		for (size_t j = 0; j < impl_->junctions.size(); ++j) { total += 1.0 / std::pow(j + 1.0, 1.2); impl_->junction_cdf[j] = total; }
		for (size_t j = 0; j < impl_->junctions.size(); ++j) impl_->junction_cdf[j] /= total;
	}
}

void Generator::write_fasta(const std::string& path) const {
	FILE* f = fopen(path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + path);
	for (size_t contig = 0; contig < contig_names_.size(); ++contig) {
		fprintf(f, ">%s synthetic\n", contig_names_[contig].c_str());
		const std::string& sequence = contig_sequences_[contig];
		for (size_t p = 0; p < sequence.size(); p += 60) {
			fwrite(sequence.data() + p, 1, std::min<size_t>(60, sequence.size() - p), f);
			fputc('\n', f);
		}
	}
	fclose(f);
}

// A blacklist and a known-fusions file (the reference's -b / -k) that exercise every kind of item: gene names, exact positions (1-based, optionally
// with a strand), ranges, contig names with a "chr" prefix or a trailing asterisk, every keyword, comments and malformed lines.  Derived from
// the junction table by a generator of its own, so that lines hit candidates: well supported junctions for the blacklist, all of them (mostly
// poorly supported ones) for the known fusions.
void Generator::write_rule_files(const std::string& blacklist_path, const std::string& known_fusions_path) const {
	Rng rng(config_.seed ^ 0xB1AC4157ULL);
	const std::vector<Junction>& junctions = impl_->junctions;
	auto position = [&](const Side& side, bool with_strand, bool chr_prefix) {
		std::string text = with_strand ? (rng.chance(0.5) ? "+" : "-") : "";
		return text + (chr_prefix ? "chr" : "") + contig_names_[side.contig] + ":" + std::to_string(side.bp + 1);
	};
	auto range_of_gene = [&](int gene, bool with_strand) {
		const Gene& g = genes_[gene];
		const int shrink = rng.range(-2000, (g.end - g.start) / 3); // sometimes wider than the gene, sometimes covering less than half of it
		std::string text = with_strand ? (g.plus ? "+" : "-") : "";
		return text + contig_names_[g.contig] + ":" + std::to_string(std::max(1, g.start + 1 + shrink)) + "-" + std::to_string(g.end + 1 - shrink / 2);
	};
	static const char* const keywords[] = { "any", "split_read_donor", "split_read_acceptor", "split_read_any", "discordant_mates", "read_through", "low_support", "filter_spliced", "not_both_spliced" };
	FILE* f = fopen(blacklist_path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + blacklist_path);
	fprintf(f, "# synthetic blacklist\n\n");
	for (size_t j = 0; j < junctions.size(); ++j) {
		const Junction& junction = junctions[j];
		if (junction.a.gene < 0 || junction.b.gene < 0 || !rng.chance(j < 40 ? 0.5 : 0.15)) continue;
		const bool swap = rng.chance(0.5);
		const Side& first = swap ? junction.b : junction.a;
		const Side& second = swap ? junction.a : junction.b;
		switch (rng.range(0, 7)) {
			case 0: fprintf(f, "%s\t%s\n", genes_[first.gene].name.c_str(), genes_[second.gene].name.c_str()); break;
			case 1: fprintf(f, "%s\t%s\n", position(first, rng.chance(0.6), rng.chance(0.3)).c_str(), position(second, rng.chance(0.4), false).c_str()); break;
			case 2: fprintf(f, "%s\t%s\n", range_of_gene(first.gene, rng.chance(0.6)).c_str(), range_of_gene(second.gene, rng.chance(0.3)).c_str()); break;
			case 3: fprintf(f, "%s\t%s\n", genes_[first.gene].name.c_str(), keywords[rng.below(9)]); break;
			case 4: fprintf(f, "%s\t%s\n", position(first, false, false).c_str(), keywords[rng.below(9)]); break;
			case 5: fprintf(f, "%s\t%s\n", range_of_gene(first.gene, false).c_str(), keywords[rng.below(9)]); break;
			case 6: fprintf(f, "%s\t%s\n", genes_[first.gene].name.c_str(), position(second, false, false).c_str()); break;
			default: { // discordant mates near a position: shifted breakpoints
				Side shifted = second; shifted.bp += rng.range(-300, 300);
				fprintf(f, "%s\t%s\r\n", genes_[first.gene].name.c_str(), position(shifted, false, false).c_str()); // DOS line end
			}
		}
	}
	// contigs by prefix, and lines the parser has to skip
	fprintf(f, "%s*:1-%d\tdiscordant_mates\n", contig_names_[0].c_str(), 30000);
	fprintf(f, "NC_*:1-7900\tany\n");
	fprintf(f, "NOT_A_GENE\tany\n%s\n%s\t\n1:abc\tany\n1:100-\tany\n1:-5\tany\nchrUn_1:5\tany\n%s\tno_such_keyword\n\tany\n1: 5\tany\n",
	        genes_[0].name.c_str(), genes_[0].name.c_str(), genes_[0].name.c_str());
	fclose(f);

	f = fopen(known_fusions_path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + known_fusions_path);
	fprintf(f, "# synthetic known fusions\n");
	for (size_t j = 0; j < junctions.size(); ++j) {
		const Junction& junction = junctions[j];
		if (junction.a.gene < 0 || junction.b.gene < 0 || !rng.chance(0.6)) continue;
		const bool swap = rng.chance(0.3); // the junction table holds the 5' gene first: a swapped line only matches an ambiguous transcript start
		const Side& first = swap ? junction.b : junction.a;
		const Side& second = swap ? junction.a : junction.b;
		switch (rng.range(0, 3)) {
			case 0: case 1: fprintf(f, "%s\t%s\n", genes_[first.gene].name.c_str(), genes_[second.gene].name.c_str()); break;
			case 2: fprintf(f, "%s\t%s\n", position(first, false, rng.chance(0.3)).c_str(), position(second, false, false).c_str()); break;
			default: fprintf(f, "%s\t%s\n", range_of_gene(first.gene, false).c_str(), genes_[second.gene].name.c_str()); break;
		}
	}
	fprintf(f, "%s\tany\nNOT_A_GENE\t%s\n", genes_[0].name.c_str(), genes_[0].name.c_str()); // keywords are not allowed here: "any" is an unknown gene
	fclose(f);

	// tags (-t): like the known fusions with a third column, some tags with characters that the output format reserves
	const std::string tags_path = known_fusions_path.substr(0, known_fusions_path.size() - std::string("known_fusions.tsv").size()) + "tags.tsv";
	f = fopen(tags_path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + tags_path);
	fprintf(f, "# synthetic tags\n");
	static const char* const tag_names[] = { "Mitelman", "cancer gene, curated", "COSMIC", "known pair", "recurrent_artifact", "ChimerDB 4.0" };
	for (size_t j = 0; j < junctions.size(); ++j) {
		const Junction& junction = junctions[j];
		if (junction.a.gene < 0 || junction.b.gene < 0 || !rng.chance(0.5)) continue;
		const bool swap = rng.chance(0.2);
		const Side& first = swap ? junction.b : junction.a;
		const Side& second = swap ? junction.a : junction.b;
		const char* tag = tag_names[rng.below(6)];
		switch (rng.range(0, 3)) {
			case 0: case 1: fprintf(f, "%s\t%s\t%s\n", genes_[first.gene].name.c_str(), genes_[second.gene].name.c_str(), tag); break;
			case 2: fprintf(f, "%s\t%s\t%s\n", position(first, false, false).c_str(), genes_[second.gene].name.c_str(), tag); break;
			default: fprintf(f, "%s\t%s\t%s\n", range_of_gene(first.gene, false).c_str(), range_of_gene(second.gene, false).c_str(), tag); break;
		}
	}
	fprintf(f, "%s\t%s\n%s\tNOT_A_GENE\ttag\n", genes_[0].name.c_str(), genes_[1].name.c_str(), genes_[0].name.c_str()); // no tag; unknown gene
	fclose(f);

	// protein domains (-p, GFF3): two to four domains inside the coding region of every second coding gene, some with a name of several
	// domains (added up by the reference), percent-encoded characters, and lines the parser skips
	const std::string domains_path = known_fusions_path.substr(0, known_fusions_path.size() - std::string("known_fusions.tsv").size()) + "protein_domains.gff3";
	f = fopen(domains_path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + domains_path);
	fprintf(f, "##gff-version 3\n");
	static const char* const domain_names[] = { "Protein_kinase_domain", "SH2 domain", "Zinc finger%2C C2H2", "Ig-like|V-type", "PDZ", "Helix-loop-helix%20motif" };
	for (size_t g = 0; g < genes_.size(); ++g) {
		const Gene& gene = genes_[g];
		const Transcript& t = gene.transcripts[0];
		if (t.cds_start < 0 || !rng.chance(0.5)) continue;
		const int n_domains = rng.range(2, 4);
		for (int k = 0; k < n_domains; ++k) {
			const Exon& exon = t.exons[rng.below(t.exons.size())];
			const int start = std::max(exon.start, t.cds_start), end = std::min(exon.end, t.cds_end);
			if (end - start < 30) continue;
			const int from = rng.range(start, end - 20), to = rng.range(from + 10, end);
			const bool by_name_only = rng.chance(0.2); // an id the annotation does not know: the gene is found by its name
			fprintf(f, "%s%s\tsynth\tprotein_domain\t%d\t%d\t.\t%c\t.\tName=%s;gene_name=%s;gene_id=%s;color=#808080\n", rng.chance(0.3) ? "chr" : "", contig_names_[gene.contig].c_str(), from + 1, to + 1, gene.plus ? '+' : '-',
			        domain_names[rng.below(6)], gene.name.c_str(), by_name_only ? "ENSG99999999999.1" : gene.id.c_str());
		}
	}
	fprintf(f, "1\tsynth\tprotein_domain\t10\t20\t.\t+\t.\tName=Orphan;gene_name=NOT_A_GENE;gene_id=ENSG99999999998.1\n1\tsynth\tprotein_domain\tx\t20\t.\t+\t.\tName=Bad;gene_name=%s;gene_id=%s\n"
	           "1\tsynth\tprotein_domain\t10\t20\t.\t+\t.\tgene_name=%s;gene_id=%s\nchrNowhere\tsynth\tprotein_domain\t10\t20\t.\t+\t.\tName=Lost;gene_name=%s;gene_id=%s\n",
	        genes_[0].name.c_str(), genes_[0].id.c_str(), genes_[0].name.c_str(), genes_[0].id.c_str(), genes_[0].name.c_str(), genes_[0].id.c_str());
	fclose(f);

	// structural variants from WGS (-d): genomic breakpoints close to the junctions, in Arriba's four-column format and as VCF records (BND, DEL, DUP,
	// INV), some of them not usable (filter not PASS, single breakends, malformed lines)
	const std::string variants_path = known_fusions_path.substr(0, known_fusions_path.size() - std::string("known_fusions.tsv").size()) + "sv.tsv";
	f = fopen(variants_path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + variants_path);
	fprintf(f, "##fileformat=VCFv4.2 (mixed with Arriba's own format)\n");
	for (size_t j = 0; j < junctions.size(); ++j) {
		const Junction& junction = junctions[j];
		if (!rng.chance(0.6)) continue;
		for (int variant = 0; variant < 2; ++variant) { // the directions as the junction table has them, and inverted: one of the two is how the caller sees the event
			const bool upstream_a = variant == 0 ? junction.a.upstream : !junction.a.upstream, upstream_b = variant == 0 ? junction.b.upstream : !junction.b.upstream;
			const int spread = rng.chance(0.2) ? 150000 : 3000; // some farther away than -D allows
			const int position_a = std::max(1, junction.a.bp + 1 + (upstream_a ? -rng.range(0, spread) + rng.range(0, 7) : rng.range(0, spread) - rng.range(0, 7)));
			const int position_b = std::max(1, junction.b.bp + 1 + (upstream_b ? -rng.range(0, spread) + rng.range(0, 7) : rng.range(0, spread) - rng.range(0, 7)));
			const std::string contig_a = contig_names_[junction.a.contig], contig_b = contig_names_[junction.b.contig];
			switch (rng.range(0, 5)) {
				case 0: case 1: fprintf(f, "%s:%d\t%s:%d\t%s\t%s\n", contig_a.c_str(), position_a, contig_b.c_str(), position_b, upstream_a ? "upstream" : "downstream", upstream_b ? "upstream" : "downstream"); break;
				case 2: fprintf(f, "chr%s:%d\t%s:%d\t%s\t%s\n", contig_a.c_str(), position_a, contig_b.c_str(), position_b, upstream_a ? "-" : "+", upstream_b ? "-" : "+"); break;
				case 3: { // VCF breakend
					const char bracket = upstream_b ? '[' : ']';
					const std::string mate = contig_b + ":" + std::to_string(position_b);
					const std::string alt = upstream_a ? std::string(1, bracket) + mate + bracket + "N" : "N" + std::string(1, bracket) + mate + bracket;
					fprintf(f, "%s\t%d\tbnd_%zu\tN\t%s\t.\t%s\tSVTYPE=BND;MATEID=x\n", contig_a.c_str(), position_a, j, alt.c_str(), rng.chance(0.85) ? "PASS" : "LowQual");
					break;
				}
				case 4: if (junction.a.contig == junction.b.contig) { // VCF with END: the type sets the directions
					static const char* const types[] = { "DEL", "DUP", "INV" };
					fprintf(f, "%s\t%d\tsv_%zu\tN\t<%s>\t.\tPASS\tIMPRECISE;SVTYPE=%s;END=%d\n", contig_a.c_str(), std::min(position_a, position_b), j, types[j % 3], types[j % 3], std::max(position_a, position_b));
					break;
				} // else: fall through to a breakend without a mate
				default: fprintf(f, "%s\t%d\tsingle_%zu\tN\tN.\t.\tPASS\tSVTYPE=BND\n", contig_a.c_str(), position_a, j); break;
			}
		}
	}
	fprintf(f, "1:100\t2:200\tsideways\tupstream\nnowhere:5\t1:7\tupstream\tupstream\n1\t500\tx\tN\t<CNV>\t.\tPASS\tSVTYPE=CNV;END=900\n1\t500\tx\tN\tN[[\t.\tPASS\tSVTYPE=BND\nnot a variant at all\n");
	fclose(f);
}

void Generator::write_gtf(const std::string& path) const {
	FILE* f = fopen(path.c_str(), "w");
	if (f == NULL) throw std::runtime_error("cannot write " + path);
	fprintf(f, "##description: synthetic annotation\n");
	for (size_t g = 0; g < genes_.size(); ++g) {
		const Gene& gene = genes_[g];
		const char* contig = contig_names_[gene.contig].c_str();
		char strand = gene.plus ? '+' : '-';
		bool coding = gene.transcripts[0].cds_start >= 0;
		std::string gene_attributes = "gene_id \"" + gene.id + "\"; gene_type \"" + (coding ? "protein_coding" : "lncRNA") + "\"; gene_name \"" + gene.name + "\";";
		fprintf(f, "%s\tsynth\tgene\t%d\t%d\t.\t%c\t.\t%s\n", contig, gene.start + 1, gene.end + 1, strand, gene_attributes.c_str());
		for (size_t t = 0; t < gene.transcripts.size(); ++t) {
			const Transcript& transcript = gene.transcripts[t];
			std::string attributes = "gene_id \"" + gene.id + "\"; transcript_id \"" + transcript.id + "\"; gene_type \"" + (coding ? "protein_coding" : "lncRNA") + "\"; gene_name \"" + gene.name + "\";";
			fprintf(f, "%s\tsynth\ttranscript\t%d\t%d\t.\t%c\t.\t%s\n", contig, transcript.exons.front().start + 1, transcript.exons.back().end + 1, strand, attributes.c_str());
			for (size_t i = 0; i < transcript.exons.size(); ++i) {
				const Exon& exon = transcript.exons[gene.plus ? i : transcript.exons.size() - 1 - i]; // transcription order
				fprintf(f, "%s\tsynth\texon\t%d\t%d\t.\t%c\t.\t%s exon_number %d;\n", contig, exon.start + 1, exon.end + 1, strand, attributes.c_str(), (int) i + 1);
				if (transcript.cds_start >= 0) {
					int cds_start = std::max(exon.start, transcript.cds_start);
					int cds_end = std::min(exon.end, transcript.cds_end);
					if (cds_start <= cds_end)
						fprintf(f, "%s\tsynth\tCDS\t%d\t%d\t.\t%c\t0\t%s exon_number %d;\n", contig, cds_start + 1, cds_end + 1, strand, attributes.c_str(), (int) i + 1);
				}
			}
		}
	}
	fclose(f);
}

// ---------------------------------------------------------------------------------------------
// alignments
// ---------------------------------------------------------------------------------------------

namespace {

struct Builder {
	const Config& c;
	const std::vector<std::string>& sequences;
	const std::vector<Gene>& genes;
	Generator::Impl& impl;
	Rng& rng;

	// n bases starting at genomic position p moving right/left along the transcript (if p lies in one of its exons), after skipping `skip` bases
	Aln build(int contig, int gene, int p, bool to_right, int skip, int n, bool may_have_indel = false) const {
		std::vector<Exon> segments;
		const Transcript* transcript = (gene >= 0) ? &genes[gene].transcripts[0] : NULL;
		int exon_index = -1;
		if (transcript != NULL)
			for (size_t e = 0; e < transcript->exons.size(); ++e)
				if (transcript->exons[e].start <= p && p <= transcript->exons[e].end)
					exon_index = e;
		int remaining = skip + n;
		int cursor = p;
		if (exon_index < 0) {
			Exon s;
			if (to_right) { s.start = cursor; s.end = cursor + remaining - 1; } else { s.start = cursor - remaining + 1; s.end = cursor; }
			segments.push_back(s);
		} else if (to_right) {
			while (remaining > 0) {
				Exon s;
				s.start = cursor;
				s.end = std::min(transcript->exons[exon_index].end, cursor + remaining - 1);
				remaining -= s.end - s.start + 1;
				segments.push_back(s);
				if (remaining > 0) {
					if (exon_index + 1 < (int) transcript->exons.size()) {
						cursor = transcript->exons[++exon_index].start;
					} else {
						segments.back().end += remaining;
						remaining = 0;
					}
				}
			}
		} else {
			while (remaining > 0) {
				Exon s;
				s.end = cursor;
				s.start = std::max(transcript->exons[exon_index].start, cursor - remaining + 1);
				remaining -= s.end - s.start + 1;
				segments.push_back(s);
				if (remaining > 0) {
					if (exon_index > 0) {
						cursor = transcript->exons[--exon_index].end;
					} else {
						segments.back().start -= remaining;
						remaining = 0;
					}
				}
			}
		}
		// trim the skipped bases off the near end
		int to_skip = skip;
		while (to_skip > 0) {
			Exon& near = segments.front();
			int length = near.end - near.start + 1;
			if (length <= to_skip) {
				to_skip -= length;
				segments.erase(segments.begin());
			} else {
				if (to_right) near.start += to_skip; else near.end -= to_skip;
				to_skip = 0;
			}
		}
		if (!to_right)
			std::reverse(segments.begin(), segments.end());
		Aln aln;
		aln.contig = contig;
		int limit = sequences[contig].size();
		// keep inside the contig (shift whole alignment; only matters for tiny contigs)
		if (segments.front().start < 0) { int d = -segments.front().start; for (size_t i = 0; i < segments.size(); ++i) { segments[i].start += d; segments[i].end += d; } }
		if (segments.back().end >= limit) { int d = segments.back().end - limit + 1; for (size_t i = 0; i < segments.size(); ++i) { segments[i].start -= d; segments[i].end -= d; } }
		aln.start = segments.front().start;
		aln.end = segments.back().end;
		bool indel_pending = may_have_indel && c.frac_indels > 0 && rng.chance(c.frac_indels);
		for (size_t i = 0; i < segments.size(); ++i) {
			if (i > 0)
				aln.cigar.push_back(cig(segments[i].start - segments[i - 1].end - 1, OP_N));
			const int length = segments[i].end - segments[i].start + 1;
			if (indel_pending && length >= 40) { // the read differs from the assembly by a short insertion or deletion; the segment keeps its place on the reference
				indel_pending = false;
				const int a = rng.range(15, length - 20), k = rng.range(1, 3);
				aln.cigar.push_back(cig(a, OP_M));
				aln.seq.append(sequences[contig], segments[i].start, a);
				if (rng.chance(0.5)) {
					aln.cigar.push_back(cig(k, OP_D));
					aln.cigar.push_back(cig(length - a - k, OP_M));
					aln.seq.append(sequences[contig], segments[i].start + a + k, length - a - k);
				} else {
					aln.cigar.push_back(cig(k, OP_I));
					for (int b = 0; b < k; ++b) aln.seq.push_back("ACGT"[rng.below(4)]);
					aln.cigar.push_back(cig(length - a, OP_M));
					aln.seq.append(sequences[contig], segments[i].start + a, length - a);
				}
				continue;
			}
			aln.cigar.push_back(cig(length, OP_M));
			aln.seq.append(sequences[contig], segments[i].start, length);
		}
		return aln;
	}

	Side random_side() const {
		Side side;
		double u = rng.unif();
		if (u < 0.70) {
			int g = rng.below(genes.size());
			const Gene& gene = genes[g];
			side.contig = gene.contig;
			side.gene = g;
			const Transcript& t = gene.transcripts[0];
			if (rng.chance(0.6)) { // exonic
				const Exon& exon = t.exons[rng.below(t.exons.size())];
				side.bp = rng.range(exon.start, exon.end);
			} else {
				side.bp = rng.range(gene.start, gene.end);
			}
		} else if (u < 0.96 || impl.viral_contig < 0) {
			side.contig = rng.below(impl.n_main_contigs);
			side.gene = -1;
			side.bp = rng.range(5000, (int) sequences[side.contig].size() - 5000);
		} else if (u < 0.985) {
			side.contig = rng.chance(0.7) ? impl.viral_contig : impl.viral_contig2;
			side.gene = -1;
			side.bp = rng.range(500, (int) sequences[side.contig].size() - 500);
		} else {
			side.contig = impl.boring_contig;
			side.gene = -1;
			side.bp = rng.range(1000, (int) sequences[side.contig].size() - 1000);
		}
		side.upstream = rng.chance(0.5);
		return side;
	}

	Junction pick_junction(bool& recurrent) const {
		recurrent = !rng.chance(c.frac_noise);
		if (recurrent) {
			double u = rng.unif();
			size_t j = std::lower_bound(impl.junction_cdf.begin(), impl.junction_cdf.end(), u) - impl.junction_cdf.begin();
			if (j >= impl.junctions.size()) j = impl.junctions.size() - 1;
			return impl.junctions[j];
		}
		Junction junction;
		junction.a = random_side();
		if (rng.chance(c.frac_same_gene) && junction.a.gene >= 0) { // both sides in the same gene
			const Gene& gene = genes[junction.a.gene];
			junction.b = junction.a;
			junction.b.bp = rng.range(gene.start, gene.end);
			junction.b.upstream = rng.chance(0.5);
		} else {
			junction.b = random_side();
		}
		return junction;
	}

	// split-read triplet: [0] split read (primary, SA tag), [1] its mate, [2] supplementary
	Fragment split_read(const Junction& junction) const {
		bool mate_on_b = rng.chance(0.5);
		const Side& y = mate_on_b ? junction.b : junction.a; // side holding SPLIT_READ + MATE1
		const Side& x = mate_on_b ? junction.a : junction.b; // side holding SUPPLEMENTARY
		int L = c.read_length;
		int clip = rng.range(c.clip_min, std::min(c.clip_max, L - 20));
		int anchored = L - clip;
		Aln split = build(y.contig, y.gene, y.bp, y.upstream, 0, anchored);
		Aln supplementary = build(x.contig, x.gene, x.bp, x.upstream, 0, clip);
		int offset = rng.range(std::max(0, anchored - 60), anchored + 200);
		Aln mate = build(y.contig, y.gene, y.bp, y.upstream, offset, L, true);
		bool split_forward = y.upstream;
		bool supplementary_forward = !x.upstream;
		std::string clipped = (split_forward == supplementary_forward) ? supplementary.seq : revcomp(supplementary.seq);
		if (c.frac_clip_from_partner > 0 && rng.chance(c.frac_clip_from_partner)) {
			// mismapper stress: the clipped segment actually stems from the split read's own gene (a paralogous/nearby locus)
			int g = y.gene;
			if (g >= 0) {
				const Gene& gene = genes[g];
				const Transcript& t = gene.transcripts[0];
				const Exon& exon = t.exons[rng.below(t.exons.size())];
				Aln donor = build(gene.contig, g, rng.range(exon.start, exon.end), true, 0, clip);
				clipped = rng.chance(0.5) ? donor.seq : revcomp(donor.seq);
				for (size_t i = 0; i < clipped.size(); ++i)
					if (rng.chance(0.05)) clipped[i] = random_base(rng);
			}
		}
		// non-template bases: the same ones in every split read of the junction, between the aligned part and the clipped part (neither alignment covers them)
		std::string non_template;
		if (c.frac_non_template > 0) {
			uint64_t h = (uint64_t) (uint32_t) junction.a.bp * 0x9E3779B97F4A7C15ull ^ (uint64_t) (uint32_t) junction.b.bp * 0xC2B2AE3D27D4EB4Full;
			h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
			if ((double) (h % 10000) < c.frac_non_template * 10000) {
				const int k = 1 + (int) ((h >> 16) % 3);
				for (int b = 0; b < k; ++b) non_template.push_back("ACGT"[(h >> (20 + 2 * b)) & 3]);
			}
		}
		const int extra = (int) non_template.size();
		bool junction_read_is_read1 = rng.chance(0.5);
		if (c.stranded) { // read1 is sense to the transcript: approximate via the split read's gene strand
			bool gene_plus = (y.gene >= 0) ? genes[y.gene].plus : true;
			junction_read_is_read1 = (split_forward == gene_plus);
		}
		Fragment fragment(3);
		Record& r0 = fragment[0];
		r0.flag = F_PAIRED | F_PROPER | (split_forward ? F_MREVERSE : F_REVERSE) | (junction_read_is_read1 ? F_READ1 : F_READ2);
		r0.contig = split.contig; r0.pos = split.start; r0.sa = true;
		if (split_forward) { r0.cigar.push_back(cig(clip + extra, OP_S)); r0.cigar.insert(r0.cigar.end(), split.cigar.begin(), split.cigar.end()); r0.seq = clipped + non_template + split.seq; }
		else { r0.cigar = split.cigar; r0.cigar.push_back(cig(clip + extra, OP_S)); r0.seq = split.seq + non_template + clipped; }
		Record& r1 = fragment[1];
		r1.flag = F_PAIRED | F_PROPER | (split_forward ? F_REVERSE : F_MREVERSE) | (junction_read_is_read1 ? F_READ2 : F_READ1);
		r1.contig = mate.contig; r1.pos = mate.start; r1.cigar = mate.cigar; r1.seq = mate.seq; r1.sa = false;
		Record& r2 = fragment[2];
		r2.flag = F_PAIRED | F_SUPPLEMENTARY | (supplementary_forward ? 0 : F_REVERSE) | (split_forward ? F_MREVERSE : 0) | (junction_read_is_read1 ? F_READ1 : F_READ2);
		r2.contig = supplementary.contig; r2.pos = supplementary.start; r2.seq = supplementary.seq; r2.sa = true;
		if (supplementary_forward) { r2.cigar = supplementary.cigar; r2.cigar.push_back(cig(anchored + extra, OP_H)); }
		else { r2.cigar.push_back(cig(anchored + extra, OP_H)); r2.cigar.insert(r2.cigar.end(), supplementary.cigar.begin(), supplementary.cigar.end()); }
		return fragment;
	}

	Fragment discordant_pair(const Junction& junction) const {
		int L = c.read_length;
		Fragment fragment(2);
		bool a_first = rng.chance(0.5);
		bool first_is_read1 = rng.chance(0.5);
		for (int i = 0; i < 2; ++i) {
			const Side& side = ((i == 0) == a_first) ? junction.a : junction.b;
			int offset = rng.range(0, 180);
			Aln aln = build(side.contig, side.gene, side.bp, side.upstream, offset, L);
			bool forward = !side.upstream; // forward mate ends before a DOWNSTREAM breakpoint
			Record& r = fragment[i];
			r.contig = aln.contig; r.pos = aln.start; r.cigar = aln.cigar; r.seq = aln.seq; r.sa = false;
			r.flag = F_PAIRED | (forward ? 0 : F_REVERSE) | (((i == 0) == first_is_read1) ? F_READ1 : F_READ2);
			if (rng.chance(0.05) && L > 40) { // a few discordant mates are soft-clipped at the outer end
				int clip = rng.range(3, 15);
				std::string junk(clip, 'A');
				for (int k = 0; k < clip; ++k) junk[k] = random_base(rng);
				// replace the first/last `clip` aligned bases by a clipped segment
				if (bam_first_op_length(r.cigar.front()) > clip + 10 && forward) {
					r.cigar.front() = cig(bam_first_op_length(r.cigar.front()) - clip, OP_M);
					r.cigar.insert(r.cigar.begin(), cig(clip, OP_S));
					r.seq.replace(0, clip, junk);
					r.pos += clip;
				}
			}
		}
		if (fragment[1].flag & F_REVERSE) fragment[0].flag |= F_MREVERSE;
		if (fragment[0].flag & F_REVERSE) fragment[1].flag |= F_MREVERSE;
		return fragment;
	}
	static int bam_first_op_length(uint32_t op) { return op >> 4; }

	// read-through: proper pair whose mates lie in neighbouring genes, either via a spliced read or as plain mates
	Fragment read_through() const {
		int L = c.read_length;
		if (impl.read_through_pairs.empty()) {
			bool recurrent;
			return discordant_pair(pick_junction(recurrent));
		}
		const std::pair<int,int>& pair = impl.read_through_pairs[rng.below(impl.read_through_pairs.size())];
		const Gene& left = genes[pair.first];
		const Gene& right = genes[pair.second];
		const Transcript& tl = left.transcripts[0];
		const Transcript& tr = right.transcripts[0];
		Fragment fragment(2);
		if (rng.chance(0.6)) { // spliced read spanning the gene boundary
			const Exon& donor = tl.exons[rng.range(std::max(0, (int) tl.exons.size() - 3), tl.exons.size() - 1)];
			const Exon& acceptor = tr.exons[rng.range(0, std::min<int>(2, tr.exons.size() - 1))];
			int a = rng.range(15, L - 15);
			Aln first = build(left.contig, pair.first, donor.end, false, 0, a);
			Aln second = build(right.contig, pair.second, acceptor.start, true, 0, L - a);
			Record spliced;
			spliced.contig = left.contig; spliced.pos = first.start; spliced.sa = false;
			spliced.cigar = first.cigar;
			spliced.cigar.push_back(cig(second.start - first.end - 1, OP_N));
			spliced.cigar.insert(spliced.cigar.end(), second.cigar.begin(), second.cigar.end());
			spliced.seq = first.seq + second.seq;
			bool spliced_is_forward = rng.chance(0.5);
			Aln other = spliced_is_forward ? build(right.contig, pair.second, acceptor.start, true, rng.range(L - a, L - a + 150), L, true)
			                               : build(left.contig, pair.first, donor.end, false, rng.range(a, a + 150), L, true);
			Record mate;
			mate.contig = other.contig; mate.pos = other.start; mate.cigar = other.cigar; mate.seq = other.seq; mate.sa = false;
			bool spliced_is_read1 = rng.chance(0.5);
			spliced.flag = F_PAIRED | F_PROPER | (spliced_is_forward ? F_MREVERSE : F_REVERSE) | (spliced_is_read1 ? F_READ1 : F_READ2);
			mate.flag = F_PAIRED | F_PROPER | (spliced_is_forward ? F_REVERSE : F_MREVERSE) | (spliced_is_read1 ? F_READ2 : F_READ1);
			fragment[0] = spliced; fragment[1] = mate;
		} else { // forward mate in the left gene, reverse mate in the right gene
			const Exon& le = tl.exons.back();
			const Exon& re = tr.exons.front();
			Aln forward = build(left.contig, pair.first, le.end, false, rng.range(0, 60), L, true);
			Aln reverse = build(right.contig, pair.second, re.start, true, rng.range(0, 60), L, true);
			bool forward_is_read1 = rng.chance(0.5);
			fragment[0].contig = forward.contig; fragment[0].pos = forward.start; fragment[0].cigar = forward.cigar; fragment[0].seq = forward.seq; fragment[0].sa = false;
			fragment[0].flag = F_PAIRED | F_PROPER | F_MREVERSE | (forward_is_read1 ? F_READ1 : F_READ2);
			fragment[1].contig = reverse.contig; fragment[1].pos = reverse.start; fragment[1].cigar = reverse.cigar; fragment[1].seq = reverse.seq; fragment[1].sa = false;
			fragment[1].flag = F_PAIRED | F_PROPER | F_REVERSE | (forward_is_read1 ? F_READ2 : F_READ1);
		}
		if (rng.chance(0.5))
			std::swap(fragment[0], fragment[1]);
		return fragment;
	}

	// ordinary proper pair inside one transcript; occasionally an internal-tandem-duplication read or an adapter-clipped pair
	// a read pair over one of the recurrent internal tandem duplications: the forward read runs to the end of the duplicated segment and re-enters it
	Fragment itd_hotspot_pair() const {
		const int L = c.read_length;
		const Generator::Impl::ItdHotspot& hotspot = impl.itd_hotspots[rng.below(impl.itd_hotspots.size())];
		const Gene& gene = genes[hotspot.gene];
		const int m = rng.range(45, 65), clip = L - m, insert = rng.range(L + 20, L + 120);
		Fragment fragment(2);
		Record& f = fragment[0];
		Record& r = fragment[1];
		Aln reverse = build(gene.contig, hotspot.gene, hotspot.q + hotspot.d - m, true, insert - L, L);
		const bool forward_is_read1 = rng.chance(0.5);
		f.contig = gene.contig; f.pos = hotspot.q + hotspot.d - m; f.sa = false;
		f.cigar.push_back(cig(m, OP_M)); f.cigar.push_back(cig(clip, OP_S));
		f.seq = sequences[gene.contig].substr(hotspot.q + hotspot.d - m, m) + sequences[gene.contig].substr(hotspot.q, clip);
		r.contig = reverse.contig; r.pos = reverse.start; r.cigar = reverse.cigar; r.seq = reverse.seq; r.sa = false;
		f.flag = F_PAIRED | F_PROPER | F_MREVERSE | (forward_is_read1 ? F_READ1 : F_READ2);
		r.flag = F_PAIRED | F_PROPER | F_REVERSE | (forward_is_read1 ? F_READ2 : F_READ1);
		if (rng.chance(0.5))
			std::swap(fragment[0], fragment[1]);
		return fragment;
	}

	Fragment normal_pair() const {
		if (!impl.itd_hotspots.empty() && rng.chance(c.frac_itd_hotspot)) return itd_hotspot_pair();
		int L = c.read_length;
		double u = rng.unif();
		size_t g = std::lower_bound(impl.gene_cdf.begin(), impl.gene_cdf.end(), u) - impl.gene_cdf.begin();
		if (g >= genes.size()) g = genes.size() - 1;
		const Gene& gene = genes[g];
		const Transcript& t = gene.transcripts[0];
		Fragment fragment(2);
		const Exon& exon = t.exons[rng.below(t.exons.size())];
		int anchor = rng.range(exon.start, exon.end);
		int insert = rng.range(L + 20, L + 280);
		Aln forward = build(gene.contig, g, anchor, true, 0, L);
		Aln reverse = build(gene.contig, g, anchor, true, insert - L, L);
		bool on_viral_contig = impl.viral_contig >= 0 && rng.chance(0.01);
		if (on_viral_contig) { // viral expression
			int contig = impl.viral_contig;
			anchor = rng.range(100, (int) sequences[contig].size() - 600);
			forward = build(contig, -1, anchor, true, 0, L);
			reverse = build(contig, -1, anchor, true, insert - L, L);
		}
		bool forward_is_read1 = rng.chance(0.5);
		Record& f = fragment[0];
		Record& r = fragment[1];
		f.contig = forward.contig; f.pos = forward.start; f.cigar = forward.cigar; f.seq = forward.seq; f.sa = false;
		r.contig = reverse.contig; r.pos = reverse.start; r.cigar = reverse.cigar; r.seq = reverse.seq; r.sa = false;
		f.flag = F_PAIRED | F_PROPER | F_MREVERSE | (forward_is_read1 ? F_READ1 : F_READ2);
		r.flag = F_PAIRED | F_PROPER | F_REVERSE | (forward_is_read1 ? F_READ2 : F_READ1);
		if (!on_viral_contig && rng.chance(c.frac_itd) && exon.end - exon.start > 160) {
			// internal tandem duplication: the forward read runs to the end of the duplicated segment [q, q+d) and then re-enters it at q
			int d = rng.range(12, 60);
			int q = rng.range(exon.start + 70, exon.end - d - 10);
			int m = rng.range(40, std::min(70, q + d - exon.start));
			int clip = L - m;
			f.pos = q + d - m;
			f.cigar.clear();
			f.cigar.push_back(cig(m, OP_M));
			f.cigar.push_back(cig(clip, OP_S));
			f.seq = sequences[gene.contig].substr(q + d - m, m) + sequences[gene.contig].substr(q, clip);
		} else if (!on_viral_contig && rng.chance(0.02)) {
			// short insert: both mates start at the same position and carry adapter clips
			int clip = rng.range(5, 20);
			std::string adapter(clip, 'A');
			for (int k = 0; k < clip; ++k) adapter[k] = "AGATCGGAAGAGCACACGTC"[k % 20];
			Aln both = build(gene.contig, g, anchor, true, 0, L - clip);
			f.pos = both.start; f.cigar = both.cigar; f.cigar.push_back(cig(clip, OP_S)); f.seq = both.seq + adapter;
			r.pos = both.start; r.cigar.clear(); r.cigar.push_back(cig(clip, OP_S)); r.cigar.insert(r.cigar.end(), both.cigar.begin(), both.cigar.end()); r.seq = revcomp(adapter) + both.seq;
		}
		if (rng.chance(0.5))
			std::swap(fragment[0], fragment[1]);
		return fragment;
	}
};

// ---- BAM encoding -----------------------------------------------------------------------------

struct BamEncoder {
	std::vector<uint8_t> buffer;
	static void put32(std::vector<uint8_t>& out, uint32_t v) { out.push_back(v & 255); out.push_back(v >> 8 & 255); out.push_back(v >> 16 & 255); out.push_back(v >> 24 & 255); }
	static void put16(std::vector<uint8_t>& out, uint16_t v) { out.push_back(v & 255); out.push_back(v >> 8); }
	static uint8_t code(char base) {
		switch (base) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
	}
	void header(const std::vector<std::string>& names, const std::vector<std::string>& sequences) {
		std::string text = "@HD\tVN:1.4\tSO:unsorted\n";
		for (size_t i = 0; i < names.size(); ++i)
			text += "@SQ\tSN:" + names[i] + "\tLN:" + std::to_string(sequences[i].size()) + "\n";
		text += "@PG\tID:gen_synth\tPN:gen_synth\n";
		buffer.insert(buffer.end(), {'B', 'A', 'M', 1});
		put32(buffer, text.size());
		buffer.insert(buffer.end(), text.begin(), text.end());
		put32(buffer, names.size());
		for (size_t i = 0; i < names.size(); ++i) {
			put32(buffer, names[i].size() + 1);
			buffer.insert(buffer.end(), names[i].begin(), names[i].end());
			buffer.push_back(0);
			put32(buffer, sequences[i].size());
		}
	}
	void record(const std::string& name, const Record& r, const Record* mate) {
		size_t start = buffer.size();
		put32(buffer, 0); // block_size, patched below
		put32(buffer, r.contig);
		put32(buffer, r.pos);
		buffer.push_back(name.size() + 1);
		buffer.push_back(r.nh > 1 ? 3 : 255); // mapq
		put16(buffer, 4680);
		put16(buffer, r.cigar.size());
		put16(buffer, r.flag);
		put32(buffer, r.seq.size());
		put32(buffer, mate ? mate->contig : -1);
		put32(buffer, mate ? mate->pos : -1);
		put32(buffer, 0);
		buffer.insert(buffer.end(), name.begin(), name.end());
		buffer.push_back(0);
		for (size_t i = 0; i < r.cigar.size(); ++i)
			put32(buffer, r.cigar[i]);
		for (size_t i = 0; i < r.seq.size(); i += 2)
			buffer.push_back(code(r.seq[i]) << 4 | (i + 1 < r.seq.size() ? code(r.seq[i + 1]) : 0));
		buffer.insert(buffer.end(), r.seq.size(), 30);
		buffer.insert(buffer.end(), {'N', 'H', 'C', (uint8_t) r.nh});
		if (!r.no_hi) buffer.insert(buffer.end(), {'H', 'I', 'C', (uint8_t) r.hi});
		if (r.sa) {
			const char sa[] = "SAZ1,1,+,50M50S,255,0;";
			buffer.insert(buffer.end(), sa, sa + sizeof(sa)); // includes the terminating NUL
		}
		uint32_t block_size = buffer.size() - start - 4;
		buffer[start] = block_size & 255; buffer[start + 1] = block_size >> 8 & 255; buffer[start + 2] = block_size >> 16 & 255; buffer[start + 3] = block_size >> 24 & 255;
	}
};

}

void Generator::stream_bam(const ByteSink& sink) {
	if (genes_.empty())
		build_reference();
	if (config_.read_seed != 0) impl_->rng = Rng(config_.read_seed);
	BamEncoder encoder;
	encoder.header(contig_names_, contig_sequences_);
	sink(encoder.buffer.data(), encoder.buffer.size());
	records_written_ = 0;
	stream_records(impl_->rng, 0, config_.fragments, sink, records_written_);
}

// the records of `fragments` chimeric fragments (and the ordinary pairs between them) drawn from `rng`; names count up from `first_serial`
void Generator::stream_records(Rng& rng, uint64_t first_serial, long fragments, const ByteSink& sink, long& records_written) const {
	const Config& c = config_;
	Builder builder = { c, contig_sequences_, genes_, *impl_, rng };
	BamEncoder encoder;

	struct Pending { std::string name; Record record; bool has_mate; Record mate; };
	std::vector<Pending> pool; // delay pool for separate_mates
	std::vector<Fragment> reservoir; // recent chimeric fragments for PCR duplicates
	uint64_t serial = first_serial;

	auto apply_errors = [&](Fragment& fragment) {
		double rate = rng.chance(c.high_error_fraction) ? c.high_error_rate : c.error_rate;
		for (size_t i = 0; i < fragment.size(); ++i)
			for (size_t p = 0; p < fragment[i].seq.size(); ++p)
				if (rng.chance(rate))
					fragment[i].seq[p] = random_base(rng);
	};
	auto make_name = [&]() {
		uint64_t id = c.shuffle_names ? (serial * 2654435761ULL) % 10000000000ULL : serial;
		++serial;
		char buffer[32];
		snprintf(buffer, sizeof(buffer), "r%010llu", (unsigned long long) id);
		std::string name(buffer);
		// --name-length: as long as the names of an Illumina run ("A00123:45:HXXXXXXXX:1:1101:12345:12345": 38-45 characters); the padding follows the number, so the order of the names stays
		if ((int) name.size() < c.name_length) name += std::string(":A00123:45:HXXXXXXXX:1:1101:00000:00000:pad").substr(0, (size_t) c.name_length - name.size());
		return name;
	};
	auto flush = [&](bool everything) {
		while (!pool.empty() && (everything || pool.size() > 48)) {
			size_t pick = rng.below(pool.size());
			encoder.record(pool[pick].name, pool[pick].record, pool[pick].has_mate ? &pool[pick].mate : NULL);
			pool[pick] = pool.back();
			pool.pop_back();
		}
	};
	auto emit = [&](const std::string& name, Fragment fragment, int hi, int nh) {
		apply_errors(fragment);
		// the split read's supplementary shares its bases with the primary: keep them consistent after errors
		if (fragment.size() == 3) {
			const Record& primary = fragment[0];
			Record& supplementary = fragment[2];
			uint32_t clip = (primary.cigar.front() & 15) == OP_S ? primary.cigar.front() >> 4 : primary.cigar.back() >> 4;
			std::string clipped = (primary.cigar.front() & 15) == OP_S ? primary.seq.substr(0, clip) : primary.seq.substr(primary.seq.size() - clip);
			bool same_strand = ((primary.flag ^ supplementary.flag) & F_REVERSE) == 0;
			supplementary.seq = same_strand ? clipped : revcomp(clipped);
			if (c.soft_clip_supplementary) { // the supplementary alignment carries the whole read, its unaligned part soft-clipped
				for (size_t k = 0; k < supplementary.cigar.size(); ++k)
					if ((supplementary.cigar[k] & 15) == OP_H) supplementary.cigar[k] = cig(supplementary.cigar[k] >> 4, OP_S);
				supplementary.seq = same_strand ? primary.seq : revcomp(primary.seq);
			}
		}
		if (c.frac_n_bases > 0) {
			for (size_t i = 0; i < fragment.size(); ++i)
				if (!(fragment.size() == 3 && i == 2)) // (the supplementary alignment takes the bases of its primary)
					for (size_t p = 0; p < fragment[i].seq.size(); ++p)
						if (rng.chance(c.frac_n_bases)) fragment[i].seq[p] = 'N';
			if (fragment.size() == 3) {
				const Record& primary = fragment[0]; Record& supplementary = fragment[2];
				const bool same_strand = ((primary.flag ^ supplementary.flag) & F_REVERSE) == 0;
				if (c.soft_clip_supplementary) supplementary.seq = same_strand ? primary.seq : revcomp(primary.seq);
				else {
					uint32_t clip = (primary.cigar.front() & 15) == OP_S ? primary.cigar.front() >> 4 : primary.cigar.back() >> 4;
					// (with non-template bases the clipped part of the primary is longer than the supplementary alignment: its far end is what the supplementary holds)
					std::string clipped = (primary.cigar.front() & 15) == OP_S ? primary.seq.substr(0, clip) : primary.seq.substr(primary.seq.size() - clip);
					const size_t aligned = supplementary.seq.size();
					if (clipped.size() >= aligned) clipped = (primary.cigar.front() & 15) == OP_S ? clipped.substr(0, aligned) : clipped.substr(clipped.size() - aligned);
					supplementary.seq = same_strand ? clipped : revcomp(clipped);
				}
			}
		}
		if (c.single_end) { // the records of the read that the first record belongs to, without the flags of a paired library
			Fragment kept;
			const uint16_t read = fragment[0].flag & (F_READ1 | F_READ2);
			for (size_t i = 0; i < fragment.size(); ++i)
				if ((fragment[i].flag & (F_READ1 | F_READ2)) == read) { kept.push_back(fragment[i]); kept.back().flag &= (uint16_t) ~(F_PAIRED | F_PROPER | F_MREVERSE | F_READ1 | F_READ2); }
			fragment.swap(kept);
		}
		for (size_t i = 0; i < fragment.size(); ++i) {
			fragment[i].hi = hi;
			fragment[i].nh = nh;
		}
		// a few fragments arrive in unusual record order (supplementary first)
		if (fragment.size() == 3 && rng.chance(0.3))
			std::swap(fragment[0], fragment[2]);
		for (size_t i = 0; i < fragment.size(); ++i) {
			const Record* mate = NULL;
			for (size_t k = 0; k < fragment.size(); ++k)
				if (k != i && !(fragment[k].flag & F_SUPPLEMENTARY) && ((fragment[k].flag ^ fragment[i].flag) & (F_READ1 | F_READ2)))
					mate = &fragment[k];
			if (c.separate_mates) {
				Pending pending;
				pending.name = name; pending.record = fragment[i]; pending.has_mate = mate != NULL;
				if (mate) pending.mate = *mate;
				pool.push_back(pending);
			} else {
				encoder.record(name, fragment[i], mate);
			}
			++records_written;
		}
		if (c.separate_mates)
			flush(false);
		if (encoder.buffer.size() > (4u << 20)) {
			sink(encoder.buffer.data(), encoder.buffer.size());
			encoder.buffer.clear();
		}
	};

	double normal_debt = 0;
	for (long n = 0; n < fragments; ++n) {
		// ordinary pairs interleaved with the chimeric fragments
		normal_debt += c.normal_multiplier;
		while (normal_debt >= 1) {
			emit(make_name(), builder.normal_pair(), 1, 1);
			normal_debt -= 1;
		}

		Fragment fragment;
		double u = rng.unif();
		if (!reservoir.empty() && rng.chance(c.frac_duplicates)) {
			fragment = reservoir[rng.below(reservoir.size())];
		} else {
			bool recurrent;
			if (u < c.frac_split) fragment = builder.split_read(builder.pick_junction(recurrent));
			else if (u < c.frac_split + c.frac_discordant) fragment = builder.discordant_pair(builder.pick_junction(recurrent));
			else fragment = builder.read_through();
			if (reservoir.size() < 64) reservoir.push_back(fragment); else reservoir[rng.below(64)] = fragment;
		}
		if (rng.chance(c.frac_malformed)) { // drop or corrupt a record
			if (fragment.size() == 3 && rng.chance(0.5)) fragment.pop_back();
			else if (fragment.size() == 3) fragment[2].flag &= ~F_SUPPLEMENTARY, fragment[2].flag |= F_SECONDARY;
		}
		std::string name = make_name();
		if (rng.chance(c.frac_multimappers)) {
			int copies = rng.range(2, 3);
			const bool missing_hi = c.frac_missing_hi > 0 && rng.chance(c.frac_missing_hi);
			emit(name, fragment, 1, copies);
			for (int copy = 2; copy <= copies; ++copy) {
				bool recurrent;
				Fragment other = (fragment.size() == 3) ? builder.split_read(builder.pick_junction(recurrent)) : builder.discordant_pair(builder.pick_junction(recurrent));
				for (size_t i = 0; i < other.size(); ++i) {
					if (!(other[i].flag & F_SUPPLEMENTARY))
						other[i].flag |= F_SECONDARY;
					other[i].no_hi = missing_hi;
				}
				emit(name, other, copy, copies);
			}
		} else {
			emit(name, fragment, 1, 1);
		}
	}
	flush(true);
	if (!encoder.buffer.empty())
		sink(encoder.buffer.data(), encoder.buffer.size());
}

// One BGZF block (gzip member with a 'BC' extra field holding BSIZE, payload as a single stored deflate block) appended to `out`
static void append_stored_block(std::vector<uint8_t>& out, const uint8_t* data, size_t length) {
	uint8_t header[18] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0 };
	const uint16_t bsize = (uint16_t) (18 + 5 + length + 8 - 1);
	header[16] = bsize & 255; header[17] = bsize >> 8;
	out.insert(out.end(), header, header + 18);
	const uint8_t stored[5] = { 1, (uint8_t) (length & 255), (uint8_t) (length >> 8), (uint8_t) (~length & 255), (uint8_t) ((~length >> 8) & 255) };
	out.insert(out.end(), stored, stored + 5);
	out.insert(out.end(), data, data + length);
	const uint32_t crc = crc32(crc32(0L, Z_NULL, 0), data, length);
	const uint8_t trailer[8] = { (uint8_t) (crc & 255), (uint8_t) (crc >> 8 & 255), (uint8_t) (crc >> 16 & 255), (uint8_t) (crc >> 24 & 255), (uint8_t) (length & 255), (uint8_t) (length >> 8 & 255), 0, 0 };
	out.insert(out.end(), trailer, trailer + 8);
}

// ... or with its payload deflated at `level` (what STAR writes by default, and samtools): one raw DEFLATE stream per block
static void append_deflated_block(std::vector<uint8_t>& out, const uint8_t* data, size_t length, int level) {
	z_stream z; memset(&z, 0, sizeof(z));
	if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
	uint8_t packed[65536 + 1024];
	z.next_in = (Bytef*) data; z.avail_in = (uInt) length; z.next_out = packed; z.avail_out = sizeof(packed);
	if (deflate(&z, Z_FINISH) != Z_STREAM_END) { deflateEnd(&z); throw std::runtime_error("deflate failed"); }
	const size_t size = z.total_out;
	deflateEnd(&z);
	uint8_t header[18] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0 };
	const uint16_t bsize = (uint16_t) (18 + size + 8 - 1);
	header[16] = bsize & 255; header[17] = bsize >> 8;
	out.insert(out.end(), header, header + 18);
	out.insert(out.end(), packed, packed + size);
	const uint32_t crc = crc32(crc32(0L, Z_NULL, 0), data, length);
	const uint8_t trailer[8] = { (uint8_t) (crc & 255), (uint8_t) (crc >> 8 & 255), (uint8_t) (crc >> 16 & 255), (uint8_t) (crc >> 24 & 255), (uint8_t) (length & 255), (uint8_t) (length >> 8 & 255), 0, 0 };
	out.insert(out.end(), trailer, trailer + 8);
}

// The same kind of sample written by several threads (the bench's 10^7..10^8 fragments: one thread makes ~0.2 M fragments/s).  The fragments are cut into
// segments of SEGMENT fragments; segment k draws from its own generator (seeded by read seed and k), numbers its names from k * SEGMENT * 8 and is encoded
// -- and, for BGZF, cut into stored blocks -- on its own, so the file does not depend on the number of threads.  Not the byte stream of write_bam().
void Generator::write_bam_segmented(const std::string& path, unsigned int n_threads, bool bgzf) {
	if (genes_.empty())
		build_reference();
	const long SEGMENT = 50000;
	const size_t BLOCK = 65280;
	const long n_segments = (config_.fragments + SEGMENT - 1) / SEGMENT;
	FILE* f = fopen(path.c_str(), "wb");
	if (f == NULL) throw std::runtime_error("cannot write " + path);
	auto wrap = [&](const std::vector<uint8_t>& raw, std::vector<uint8_t>& out) {
		if (!bgzf) { out = raw; return; }
		out.clear(); out.reserve(raw.size() + raw.size() / 2000 + 64);
		const int level = config_.bgzf_level;
		for (size_t at = 0; at < raw.size(); at += BLOCK) { if (level > 0) append_deflated_block(out, raw.data() + at, std::min(BLOCK, raw.size() - at), level); else append_stored_block(out, raw.data() + at, std::min(BLOCK, raw.size() - at)); }
	};
	{
		BamEncoder encoder;
		encoder.header(contig_names_, contig_sequences_);
		std::vector<uint8_t> out;
		wrap(encoder.buffer, out);
		if (fwrite(out.data(), 1, out.size(), f) != out.size()) throw std::runtime_error("short write");
	}
	std::vector<std::vector<uint8_t> > ready(n_segments), spare; // spare: buffers the writer is done with (fresh ~30 MB vectors cost more in page faults than the records in them)
	std::vector<char> done(n_segments, 0);
	std::vector<long> records(n_segments, 0);
	std::mutex mutex; std::condition_variable changed;
	long next_segment = 0, written = 0;
	const long window = (long) n_threads * 2 + 2; // segments in flight: bounds the memory
	std::exception_ptr failure;
	std::vector<std::thread> threads;
	for (unsigned int t = 0; t < std::max(1u, n_threads); ++t)
		threads.push_back(std::thread([&] {
			std::vector<uint8_t> raw, out;
			while (true) {
				long k;
				{
					std::unique_lock<std::mutex> lock(mutex);
					changed.wait(lock, [&] { return failure || next_segment >= n_segments || next_segment < written + window; });
					if (failure || next_segment >= n_segments) return;
					k = next_segment++;
					if (out.capacity() == 0 && !spare.empty()) { out.swap(spare.back()); spare.pop_back(); }
				}
				try {
					Rng rng((config_.read_seed != 0 ? config_.read_seed : config_.seed) * 0x9E3779B97F4A7C15ULL + (uint64_t) k * 0xD1B54A32D192ED03ULL + 1);
					raw.clear();
					const long count = std::min(SEGMENT, config_.fragments - k * SEGMENT);
					stream_records(rng, (uint64_t) k * SEGMENT * 8, count, [&raw](const uint8_t* data, size_t size) { raw.insert(raw.end(), data, data + size); }, records[k]);
					wrap(raw, out);
					std::unique_lock<std::mutex> lock(mutex);
					ready[k].swap(out); done[k] = 1;
					changed.notify_all();
				} catch (...) { std::unique_lock<std::mutex> lock(mutex); failure = std::current_exception(); changed.notify_all(); return; }
			}
		}));
	records_written_ = 0;
	for (long k = 0; k < n_segments && !failure; ++k) {
		std::vector<uint8_t> out;
		{
			std::unique_lock<std::mutex> lock(mutex);
			changed.wait(lock, [&] { return failure || done[k]; });
			if (failure) break;
			out.swap(ready[k]);
		}
		if (fwrite(out.data(), 1, out.size(), f) != out.size()) { std::unique_lock<std::mutex> lock(mutex); failure = std::make_exception_ptr(std::runtime_error("short write")); changed.notify_all(); break; }
		records_written_ += records[k];
		std::unique_lock<std::mutex> lock(mutex);
		out.clear(); spare.push_back(std::vector<uint8_t>()); spare.back().swap(out);
		written = k + 1;
		changed.notify_all();
	}
	{ std::unique_lock<std::mutex> lock(mutex); changed.notify_all(); }
	for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
	if (bgzf) {
		static const uint8_t eof_marker[28] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		fwrite(eof_marker, 1, 28, f);
	}
	fclose(f);
	if (failure) std::rethrow_exception(failure);
}

void Generator::write_bam(const std::string& path) {
	FILE* f = fopen(path.c_str(), "wb");
	if (f == NULL) throw std::runtime_error("cannot write " + path);
	std::vector<uint8_t> carry;
	auto write_block = [&](const uint8_t* data, size_t length) {
		// one BGZF block = gzip member with a 'BC' extra field holding BSIZE, payload as a single stored deflate block
		uint8_t header[18] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0 };
		uint16_t bsize = (uint16_t) (18 + 5 + length + 8 - 1);
		header[16] = bsize & 255; header[17] = bsize >> 8;
		fwrite(header, 1, 18, f);
		uint8_t stored[5] = { 1, (uint8_t) (length & 255), (uint8_t) (length >> 8), (uint8_t) (~length & 255), (uint8_t) ((~length >> 8) & 255) };
		fwrite(stored, 1, 5, f);
		fwrite(data, 1, length, f);
		uint32_t crc = crc32(crc32(0L, Z_NULL, 0), data, length);
		uint8_t trailer[8] = { (uint8_t) (crc & 255), (uint8_t) (crc >> 8 & 255), (uint8_t) (crc >> 16 & 255), (uint8_t) (crc >> 24 & 255),
		                       (uint8_t) (length & 255), (uint8_t) (length >> 8 & 255), 0, 0 };
		fwrite(trailer, 1, 8, f);
	};
	const size_t BLOCK = 65280;
	stream_bam([&](const uint8_t* data, size_t length) {
		carry.insert(carry.end(), data, data + length);
		size_t done = 0;
		while (carry.size() - done >= BLOCK) {
			write_block(carry.data() + done, BLOCK);
			done += BLOCK;
		}
		carry.erase(carry.begin(), carry.begin() + done);
	});
	if (!carry.empty())
		write_block(carry.data(), carry.size());
	static const uint8_t eof_marker[28] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	fwrite(eof_marker, 1, 28, f);
	fclose(f);
}

}

#ifdef GEN_SYNTH_MAIN
static void usage() {
	fprintf(stderr,
		"usage: gen_synth --out PREFIX [--seed N] [--fragments N] [--normal-mult X] [--contigs N] [--contig-len N]\n"
		"                 [--genes-per-mb X] [--read-len N] [--junctions N] [--clip-min N] [--clip-max N]\n"
		"                 [--noise X] [--dup X] [--multimap X] [--partner-clip X] [--indels X] [--non-template X] [--missing-hi X] [--single-end] [--soft-clip-supplementary] [--n-bases X] [--shuffle] [--separate-mates]\n"
		"                 [--stranded] [--no-viral] [--reference-only] [--raw-bam-to PATH] [--threads N] [--bam-only]\n"
		"writes PREFIX.fa PREFIX.gtf PREFIX.bam\n");
}

int main(int argc, char** argv) {
	synth::Config config;
	std::string out;
	bool reference_only = false, rule_files = false, bam_only = false;
	unsigned int threads = 0; // > 0: the BAM file is written in segments by that many threads (another byte stream than the default one)
	std::string raw_bam_path;
	for (int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto value = [&]() -> const char* { if (i + 1 >= argc) { usage(); exit(1); } return argv[++i]; };
		if (a == "--out") out = value();
		else if (a == "--seed") config.seed = strtoull(value(), NULL, 10);
		else if (a == "--read-seed") config.read_seed = strtoull(value(), NULL, 10);
		else if (a == "--fragments") config.fragments = atol(value());
		else if (a == "--normal-mult") config.normal_multiplier = atof(value());
		else if (a == "--contigs") config.contigs = atoi(value());
		else if (a == "--contig-len") config.contig_length = atoi(value());
		else if (a == "--genes-per-mb") config.genes_per_mb = atof(value());
		else if (a == "--gene-stack") config.gene_stack = atoi(value());
		else if (a == "--itd-hotspots") config.itd_hotspots = atoi(value());
		else if (a == "--homolog-families") config.homolog_families = atoi(value());
		else if (a == "--rule-files") rule_files = true;
		else if (a == "--itd-hotspot-frac") config.frac_itd_hotspot = atof(value());
		else if (a == "--read-len") config.read_length = atoi(value());
		else if (a == "--junctions") config.junctions = atoi(value());
		else if (a == "--clip-min") config.clip_min = atoi(value());
		else if (a == "--clip-max") config.clip_max = atoi(value());
		else if (a == "--noise") config.frac_noise = atof(value());
		else if (a == "--dup") config.frac_duplicates = atof(value());
		else if (a == "--multimap") config.frac_multimappers = atof(value());
		else if (a == "--partner-clip") config.frac_clip_from_partner = atof(value());
		else if (a == "--indels") config.frac_indels = atof(value());
		else if (a == "--non-template") config.frac_non_template = atof(value());
		else if (a == "--missing-hi") config.frac_missing_hi = atof(value());
		else if (a == "--single-end") config.single_end = true;
		else if (a == "--soft-clip-supplementary") config.soft_clip_supplementary = true;
		else if (a == "--n-bases") config.frac_n_bases = atof(value());
		else if (a == "--bgzf-level") config.bgzf_level = atoi(value());
		else if (a == "--name-length") config.name_length = atoi(value());
		else if (a == "--shuffle") config.shuffle_names = true;
		else if (a == "--separate-mates") config.separate_mates = true;
		else if (a == "--stranded") config.stranded = true;
		else if (a == "--no-viral") config.viral = false;
		else if (a == "--reference-only") reference_only = true;
		else if (a == "--raw-bam-to") raw_bam_path = value();
		else if (a == "--threads") threads = (unsigned int) atoi(value());
		else if (a == "--bam-only") bam_only = true;
		else { usage(); return 1; }
	}
	if (out.empty()) { usage(); return 1; }
	try {
		synth::Generator generator(config);
		generator.build_reference();
		if (!raw_bam_path.empty()) { // stream the raw (un-BGZF'd) BAM records to a file or FIFO; the reference files are not rewritten
			FILE* raw = fopen(raw_bam_path.c_str(), "wb");
			if (raw == NULL) throw std::runtime_error("cannot write " + raw_bam_path);
			generator.stream_bam([&](const uint8_t* data, size_t size) { if (fwrite(data, 1, size, raw) != size) throw std::runtime_error("short write"); });
			fclose(raw);
		} else {
			if (!bam_only) {
				generator.write_fasta(out + ".fa");
				generator.write_gtf(out + ".gtf");
				if (rule_files) generator.write_rule_files(out + ".blacklist.tsv", out + ".known_fusions.tsv");
			}
			if (!reference_only) {
				if (threads > 0) generator.write_bam_segmented(out + ".bam", threads, true);
				else generator.write_bam(out + ".bam");
			}
		}
		fprintf(stderr, "gen_synth: %zu contigs, %zu genes, %ld records\n", generator.contig_names().size(), generator.genes().size(), generator.records_written());
	} catch (const std::exception& e) {
		fprintf(stderr, "gen_synth: %s\n", e.what());
		return 1;
	}
	return 0;
}
#endif
