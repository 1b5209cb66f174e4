// arriba_amd/csrc/host/reference_data.cpp -- FASTA / GTF loaders, the flattened interval index and
// the host-side index queries.  Behaviour follows the reference functions cited at each definition.
#include "arriba_host.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <tuple>
#include <zlib.h>

namespace arriba {

const char* const FILTER_NAMES[FILTER_COUNT] = {
	"", "duplicates", "inconsistently_clipped", "homopolymer", "read_through", "same_gene", "small_insert_size", "long_gap", "hairpin",
	"multimappers", "mismatches", "mismappers", "relative_support", "intronic", "non_coding_neighbors", "intragenic_exonic",
	"internal_tandem_duplication", "min_support", "known_fusions", "spliced", "blacklist", "end_to_end", "in_vitro", "merge_adjacent",
	"select_best", "marginal_read_through", "short_anchor", "no_coverage", "many_spliced", "no_genomic_support", "uninteresting_contigs",
	"viral_contigs", "top_expressed_viral_contigs", "low_coverage_viral_contigs", "genomic_support", "isoforms", "low_entropy", "homologs"
};

// reference: source/common.hpp:74-80
std::string remove_chr(std::string contig) {
	if (contig.compare(0, 3, "chr") == 0 && contig.size() >= 3)
		contig = contig.substr(3);
	if (contig == "M")
		contig = "MT";
	return contig;
}

// reference: source/common.hpp:82-110 -- glob-like patterns separated by whitespace, '*' wildcard
bool is_interesting_contig(std::string contig, const std::string& interesting_contigs) {
	contig = remove_chr(contig);
	std::istringstream patterns(interesting_contigs);
	while (patterns) {
		std::string pattern;
		patterns >> pattern;
		pattern = remove_chr(pattern);
		if (pattern.empty())
			continue;
		bool is_prefix = pattern[pattern.size() - 1] == '*';
		bool is_suffix = pattern[0] == '*';
		std::replace(pattern.begin(), pattern.end(), '*', ' ');
		std::istringstream segments(pattern);
		size_t position = 0;
		while (segments) {
			std::string segment;
			segments >> segment;
			if (position == 0 && !is_suffix && contig.substr(0, segment.size()) != segment)
				break;
			if (segment.empty() && (position == contig.size() || is_prefix))
				return true;
			position = contig.find(segment, position);
			if (position == std::string::npos)
				break;
			position += segment.size();
		}
	}
	return false;
}

contig_t Contigs::add(const std::string& original_name) {
	std::string name = remove_chr(original_name);
	std::pair<std::map<std::string, contig_t>::iterator, bool> inserted = by_name.insert(std::make_pair(name, (contig_t) by_name.size()));
	if (by_name.size() == USHRT_MAX - 1)
		throw std::runtime_error("too many contigs");
	if (original_names.size() < by_name.size())
		original_names.resize(by_name.size());
	original_names[inserted.first->second] = original_name; // the reference overwrites the stored name at every call site
	return inserted.first->second;
}

// ---- line reader over zlib (reads plain, gzip and BGZF files alike) -----------------------------

namespace {

class LineReader {
public:
	explicit LineReader(const std::string& path): fill_(0), position_(0), eof_(false) {
		file_ = gzopen(path.c_str(), "rb");
		if (file_ == NULL)
			throw std::runtime_error("failed to open file: " + path);
		gzbuffer(file_, 1u << 20);
		buffer_.resize(1u << 20);
	}
	~LineReader() { gzclose(file_); }
	bool getline(std::string& line) {
		line.clear();
		while (true) {
			if (position_ == fill_) {
				if (eof_)
					break;
				int got = gzread(file_, &buffer_[0], buffer_.size());
				if (got < 0)
					throw std::runtime_error("failed to decompress file");
				if (got == 0) { eof_ = true; break; }
				fill_ = got;
				position_ = 0;
			}
			const char* start = &buffer_[position_];
			const char* newline = (const char*) memchr(start, '\n', fill_ - position_);
			if (newline != NULL) {
				line.append(start, newline - start);
				position_ += newline - start + 1;
				if (!line.empty() && line[line.size() - 1] == '\r')
					line.resize(line.size() - 1);
				return true;
			}
			line.append(start, fill_ - position_);
			position_ = fill_;
		}
		if (line.empty())
			return false;
		if (line[line.size() - 1] == '\r')
			line.resize(line.size() - 1);
		return true;
	}
private:
	gzFile file_;
	std::vector<char> buffer_;
	size_t fill_, position_;
	bool eof_;
};

// field splitter with the failure semantics of the reference's tsv_stream_t (source/read_compressed_file.cpp:64-89)
struct TsvStream {
	const std::string& data; char delimiter; size_t position; bool failed;
	TsvStream(const std::string& s, char d = '\t'): data(s), delimiter(d), position(0), failed(false) {}
	void next(std::string& out) {
		if (position >= data.size()) { failed = true; return; }
		size_t start = position;
		position = data.find(delimiter, start);
		out = data.substr(start, position - start);
		if (position < data.size()) position++;
	}
	void next(int& out) {
		if (position >= data.size()) { failed = true; return; }
		size_t start = position;
		position = data.find(delimiter, start);
		std::string field = data.substr(start, position - start);
		char* end_of_parsing;
		long int result = strtol(field.c_str(), &end_of_parsing, 10);
		out = result;
		if (!(field.c_str()[0] != ' ' && end_of_parsing != field.c_str() && *end_of_parsing == '\0' && result != LONG_MAX && result != LONG_MIN))
			failed = true;
		if (position < data.size()) position++;
	}
};

}

// reference: source/assembly.cpp:28-58
void load_assembly(Assembly& assembly, const std::string& fasta_path, Contigs& contigs, const std::string& interesting_contigs) {
	LineReader fasta(fasta_path);
	std::string line;
	int current_contig = -1;
	while (fasta.getline(line)) {
		if (line.empty())
			continue;
		if (line[0] == '>') {
			std::istringstream header(line.substr(1));
			std::string contig_name;
			header >> contig_name;
			current_contig = contigs.add(contig_name);
			if (!is_interesting_contig(contig_name, interesting_contigs))
				current_contig = -1; // skip uninteresting contigs
			else if (assembly.sequence.size() <= (size_t) current_contig)
				assembly.sequence.resize(current_contig + 1);
		} else if (current_contig >= 0) {
			for (size_t i = 0; i < line.size(); ++i)
				line[i] = toupper((unsigned char) line[i]);
			assembly.sequence[current_contig] += line;
		}
	}
	if (assembly.sequence.size() < contigs.size())
		assembly.sequence.resize(contigs.size());
}

// ---- GTF ----------------------------------------------------------------------------------------

namespace {

struct GtfFeatures { std::vector<std::string> gene_name, gene_id, transcript_id, feature_exon, feature_cds; };

void split_string(const std::string& unsplit, char separator, std::vector<std::string>& split) {
	std::stringstream ss(unsplit);
	std::string value;
	while (std::getline(ss, value, separator))
		if (!value.empty())
			split.push_back(value);
}

// reference: source/annotation.cpp:28-61
bool parse_gtf_features(std::string features, GtfFeatures& out) {
	std::replace(features.begin(), features.end(), ',', ' ');
	std::istringstream iss(features);
	while (iss) {
		std::string pair;
		iss >> pair;
		TsvStream tsv(pair, '=');
		std::string feature, value;
		tsv.next(feature);
		tsv.next(value);
		if (feature != "" && value == "") return false;
		if (feature == "gene_name") split_string(value, '|', out.gene_name);
		else if (feature == "gene_id") split_string(value, '|', out.gene_id);
		else if (feature == "transcript_id") split_string(value, '|', out.transcript_id);
		else if (feature == "feature_exon") split_string(value, '|', out.feature_exon);
		else if (feature == "feature_CDS") split_string(value, '|', out.feature_cds);
		else if (feature == "") {}
		else return false;
	}
	return !out.gene_name.empty() && !out.gene_id.empty() && !out.transcript_id.empty() && !out.feature_exon.empty() && !out.feature_cds.empty();
}

// reference: source/annotation.cpp:113-148
bool get_gtf_attribute(const std::string& attributes, const std::vector<std::string>& names, std::string& value) {
	size_t start = std::string::npos;
	for (size_t n = 0; n < names.size() && start >= attributes.size(); ++n)
		start = attributes.find(names[n] + " \"");
	if (start < attributes.size())
		start = attributes.find('"', start);
	size_t end = std::string::npos;
	if (start < attributes.size()) {
		start++;
		end = attributes.find('"', start);
	}
	if (start >= attributes.size() || end >= attributes.size()) {
		std::cerr << "WARNING: failed to extract ";
		for (size_t n = 0; n < names.size(); ++n)
			std::cerr << (n ? "|" : "") << names[n];
		std::cerr << " from line in GTF file: " << attributes << std::endl;
		return false;
	}
	value = attributes.substr(start, end - start);
	return true;
}

// reference: source/annotation.hpp:27-33
std::string strip_ensembl_version_number(const std::string& identifier) {
	std::string::size_type trim_position = std::string::npos;
	if (identifier.substr(0, 3) == "ENS" && (trim_position = identifier.find_last_of('.')) < identifier.size())
		return identifier.substr(0, trim_position);
	return identifier;
}

typedef std::tuple<std::string, contig_t, bool> feature_key_t;

}

// reference: source/annotation.cpp:161-377
void read_annotation_gtf(Annotation& annotation, const std::string& gtf_path, const std::string& gtf_features_string, Contigs& contigs, const Assembly& assembly) {
	GtfFeatures gtf_features;
	parse_gtf_features(gtf_features_string, gtf_features);

	std::vector<GeneRecord> genes;
	std::vector<ExonRecord> exons;
	std::vector<TranscriptRecord> transcript_records;
	std::map<feature_key_t, int> transcripts;             // short transcript id -> transcript
	std::map<feature_key_t, int> gene_by_id;               // short gene id -> gene
	std::map<feature_key_t, std::vector<int> > exons_by_transcript_id; // full transcript id -> exons
	struct CodingRegion { bool strand; contig_t contig; position_t start, end; std::string transcript_id; };
	std::vector<CodingRegion> coding_regions;

	const int max_gene_size = 3000000;
	std::set<int> malformed_genes;
	std::vector<feature_key_t> malformed_transcripts;
	std::set<std::string> non_unique_items;
	unsigned int new_id = 0;

	LineReader gtf(gtf_path);
	std::string line;
	while (gtf.getline(line)) {
		if (line.empty() || line[0] == '#')
			continue;
		TsvStream tsv(line);
		std::string contig, strand, feature, attributes, trash, gene_name, gene_id;
		int start = 0, end = 0;
		tsv.next(contig); tsv.next(trash); tsv.next(feature); tsv.next(start); tsv.next(end); tsv.next(trash); tsv.next(strand); tsv.next(trash); tsv.next(attributes);
		if (tsv.failed || contig.empty() || feature.empty() || strand.empty()) {
			std::cerr << "WARNING: failed to parse line in GTF file: " << line << std::endl;
			continue;
		}
		if (!get_gtf_attribute(attributes, gtf_features.gene_name, gene_name) || !get_gtf_attribute(attributes, gtf_features.gene_id, gene_id))
			continue;
		std::string short_gene_id = strip_ensembl_version_number(gene_id);
		contig_t contig_id = contigs.add(contig);
		position_t record_start = start - 1, record_end = end - 1; // GTF is one-based
		bool record_strand = strand[0] == '+';

		if (std::find(gtf_features.feature_exon.begin(), gtf_features.feature_exon.end(), feature) != gtf_features.feature_exon.end()) {
			ExonRecord exon;
			exon.contig = contig_id; exon.start = record_start; exon.end = record_end; exon.strand = record_strand;
			exon.coding_region_start = -1; exon.coding_region_end = -1;
			exon.previous_exon = -1; exon.next_exon = -1;
			std::string transcript_id;
			if (!get_gtf_attribute(attributes, gtf_features.transcript_id, transcript_id))
				continue;
			std::string short_transcript_id = strip_ensembl_version_number(transcript_id);
			std::pair<std::map<feature_key_t, int>::iterator, bool> transcript = transcripts.insert(std::make_pair(feature_key_t(short_transcript_id, contig_id, record_strand), -1));
			if (transcript.second) {
				TranscriptRecord record;
				record.id = new_id++; record.name = transcript_id; record.first_exon = -1; record.last_exon = -1; record.coding_length = 0;
				transcript.first->second = transcript_records.size();
				transcript_records.push_back(record);
			}
			exon.transcript = transcript.first->second;
			std::pair<std::map<feature_key_t, int>::iterator, bool> gene = gene_by_id.insert(std::make_pair(feature_key_t(short_gene_id, contig_id, record_strand), -1));
			if (gene.second) {
				GeneRecord record;
				record.contig = contig_id; record.start = record_start; record.end = record_end; record.strand = record_strand;
				new_id++;
				record.gene_id = gene_id; record.name = gene_name; record.exonic_length = 0; record.is_dummy = false; record.is_protein_coding = false;
				gene.first->second = genes.size();
				genes.push_back(record);
			} else {
				GeneRecord& g = genes[gene.first->second];
				if (g.start > exon.start) g.start = exon.start;
				if (g.end < exon.end) g.end = exon.end;
				if (g.contig != contig_id || g.end - g.start > max_gene_size) {
					if (non_unique_items.insert(gene_id).second)
						std::cerr << "WARNING: gene ID '" << gene_id << "' appears to be non-unique and will be ignored" << std::endl;
					malformed_genes.insert(gene.first->second);
				}
			}
			const GeneRecord& g = genes[gene.first->second];
			if (assembly.has(g.contig) && (unsigned int) g.end >= assembly.sequence[g.contig].size()) {
				if (non_unique_items.insert(gene_id).second)
					std::cerr << "WARNING: gene with ID '" << gene_id << "' extends beyond end of contig and will be ignored" << std::endl;
				malformed_genes.insert(gene.first->second);
			}
			exon.gene = gene.first->second;
			exons_by_transcript_id[feature_key_t(transcript_id, contig_id, record_strand)].push_back(exons.size());
			exons.push_back(exon);
		} else if (std::find(gtf_features.feature_cds.begin(), gtf_features.feature_cds.end(), feature) != gtf_features.feature_cds.end()) {
			CodingRegion coding_region;
			coding_region.strand = record_strand; coding_region.contig = contig_id; coding_region.start = record_start; coding_region.end = record_end;
			if (!get_gtf_attribute(attributes, gtf_features.transcript_id, coding_region.transcript_id))
				continue;
			coding_regions.push_back(coding_region);
		}
	}
	if (genes.empty())
		throw std::runtime_error("failed to parse GTF file, please consider using -G");

	// map coding regions to exons (source/annotation.cpp:301-320)
	for (size_t c = 0; c < coding_regions.size(); ++c) {
		const CodingRegion& cr = coding_regions[c];
		std::map<feature_key_t, std::vector<int> >::iterator transcript = exons_by_transcript_id.find(feature_key_t(cr.transcript_id, cr.contig, cr.strand));
		if (transcript == exons_by_transcript_id.end()) {
			std::cerr << "WARNING: CDS record has unknown transcript ID: " << cr.transcript_id << std::endl;
			continue;
		}
		for (size_t e = 0; e < transcript->second.size(); ++e) {
			ExonRecord& exon = exons[transcript->second[e]];
			if (exon.start <= cr.start && exon.end >= cr.start || exon.start <= cr.end && exon.end >= cr.end || exon.start >= cr.start && exon.end <= cr.end) {
				exon.coding_region_start = std::max(cr.start, exon.start);
				exon.coding_region_end = std::min(cr.end, exon.end);
				genes[exon.gene].is_protein_coding = true;
			}
		}
	}

	// link exons of a transcript in coordinate order (source/annotation.cpp:322-329; order: contig, end, start)
	for (std::map<feature_key_t, std::vector<int> >::iterator transcript = exons_by_transcript_id.begin(); transcript != exons_by_transcript_id.end(); ++transcript) {
		std::vector<int>& list = transcript->second;
		std::stable_sort(list.begin(), list.end(), [&](int a, int b) {
			if (exons[a].contig != exons[b].contig) return exons[a].contig < exons[b].contig;
			if (exons[a].end != exons[b].end) return exons[a].end < exons[b].end;
			return exons[a].start < exons[b].start;
		});
		for (size_t e = 0; e < list.size(); ++e) {
			exons[list[e]].previous_exon = (e > 0) ? list[e - 1] : -1;
			exons[list[e]].next_exon = (e + 1 < list.size()) ? list[e + 1] : -1;
		}
	}

	// transcript boundaries and coding length (source/annotation.cpp:331-342)
	for (size_t e = 0; e < exons.size(); ++e) {
		TranscriptRecord& t = transcript_records[exons[e].transcript];
		if (t.first_exon == -1 || exons[e].start < exons[t.first_exon].start) t.first_exon = e;
		if (t.last_exon == -1 || exons[e].end > exons[t.last_exon].end) t.last_exon = e;
	}
	for (size_t e = 0; e < exons.size(); ++e)
		if (exons[e].coding_region_start != -1 && exons[e].coding_region_end != -1)
			transcript_records[exons[e].transcript].coding_length += exons[e].coding_region_end - exons[e].coding_region_start + 1;

	// known annotation errors (source/annotation.cpp:344-356) and oversized transcripts
	std::map<std::string, contig_t>::const_iterator c;
	if ((c = contigs.by_name.find("4")) != contigs.by_name.end()) malformed_transcripts.push_back(feature_key_t("ENST00000507166", c->second, true));
	if ((c = contigs.by_name.find("6")) != contigs.by_name.end()) malformed_transcripts.push_back(feature_key_t("ENST00000467125", c->second, false));
	if ((c = contigs.by_name.find("9")) != contigs.by_name.end()) {
		malformed_transcripts.push_back(feature_key_t("ENST00000404796", c->second, true));
		malformed_transcripts.push_back(feature_key_t("ENST00000577563", c->second, true));
		malformed_transcripts.push_back(feature_key_t("ENST00000580900", c->second, true));
	}
	if ((c = contigs.by_name.find("7")) != contigs.by_name.end()) malformed_transcripts.push_back(feature_key_t("ENSMUST00000124096", c->second, false));
	for (std::map<feature_key_t, int>::iterator transcript = transcripts.begin(); transcript != transcripts.end(); ++transcript) {
		const TranscriptRecord& t = transcript_records[transcript->second];
		if (exons[t.last_exon].end - exons[t.first_exon].start > max_gene_size) {
			malformed_transcripts.push_back(transcript->first);
			std::cerr << "WARNING: transcript ID '" << std::get<0>(transcript->first) << "' appears to be non-unique and will be ignored" << std::endl;
		}
	}

	std::vector<bool> exon_alive(exons.size(), true), gene_alive(genes.size(), true);
	auto remove_gene = [&](int gene) { // source/annotation.cpp:63-79
		for (size_t e = 0; e < exons.size(); ++e)
			if (exon_alive[e] && exons[e].gene == gene)
				exon_alive[e] = false;
		gene_alive[gene] = false;
	};
	for (size_t m = 0; m < malformed_transcripts.size(); ++m) { // source/annotation.cpp:81-111
		std::map<feature_key_t, int>::iterator transcript = transcripts.find(malformed_transcripts[m]);
		if (transcript == transcripts.end())
			continue;
		int gene = -1;
		for (size_t e = 0; e < exons.size(); ++e)
			if (exon_alive[e] && exons[e].transcript == transcript->second) {
				gene = exons[e].gene;
				exon_alive[e] = false;
			}
		if (gene < 0)
			continue;
		position_t new_start = -1, new_end = -1;
		for (size_t e = 0; e < exons.size(); ++e)
			if (exon_alive[e] && exons[e].gene == gene) {
				if (new_start == -1 || new_start > exons[e].start) new_start = exons[e].start;
				if (new_end == -1 || new_end < exons[e].end) new_end = exons[e].end;
			}
		if (new_start == -1) {
			remove_gene(gene);
		} else {
			genes[gene].start = new_start;
			genes[gene].end = new_end;
		}
	}
	for (std::set<int>::iterator gene = malformed_genes.begin(); gene != malformed_genes.end(); ++gene)
		if (gene_alive[*gene])
			remove_gene(*gene);

	// compact (ids follow list order)
	std::vector<int> gene_map(genes.size(), -1), exon_map(exons.size(), -1);
	for (size_t g = 0; g < genes.size(); ++g)
		if (gene_alive[g]) { gene_map[g] = annotation.genes.size(); annotation.genes.push_back(genes[g]); }
	for (size_t e = 0; e < exons.size(); ++e)
		if (exon_alive[e]) { exon_map[e] = annotation.exons.size(); annotation.exons.push_back(exons[e]); }
	for (size_t e = 0; e < annotation.exons.size(); ++e) {
		ExonRecord& exon = annotation.exons[e];
		exon.gene = gene_map[exon.gene];
		exon.previous_exon = (exon.previous_exon >= 0) ? exon_map[exon.previous_exon] : -1;
		exon.next_exon = (exon.next_exon >= 0) ? exon_map[exon.next_exon] : -1;
	}
	annotation.transcripts = transcript_records;
	for (size_t t = 0; t < annotation.transcripts.size(); ++t) {
		TranscriptRecord& transcript = annotation.transcripts[t];
		transcript.first_exon = (transcript.first_exon >= 0) ? exon_map[transcript.first_exon] : -1;
		transcript.last_exon = (transcript.last_exon >= 0) ? exon_map[transcript.last_exon] : -1;
	}
	annotation.real_genes = annotation.genes.size();
	for (size_t g = 0; g < annotation.genes.size(); ++g)
		annotation.gene_by_name[annotation.genes[g].name] = g; // later genes win (source/annotation.cpp:372-375)
}

// ---- flat index ---------------------------------------------------------------------------------

template <class Feature> void make_flat_index(const std::vector<Feature>& features, size_t n_contigs, FlatIndex& index) {
	index.contig_offset.assign(n_contigs + 1, 0);
	index.keys.clear(); index.member_offset.clear(); index.members.clear();
	std::vector<std::vector<position_t> > keys_by_contig(n_contigs);
	for (size_t f = 0; f < features.size(); ++f) {
		if (features[f].contig >= n_contigs) continue;
		keys_by_contig[features[f].contig].push_back(features[f].end);
		keys_by_contig[features[f].contig].push_back(features[f].start - 1);
	}
	for (size_t contig = 0; contig < n_contigs; ++contig) {
		std::vector<position_t>& keys = keys_by_contig[contig];
		std::sort(keys.begin(), keys.end());
		keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
		index.contig_offset[contig] = index.keys.size();
		index.keys.insert(index.keys.end(), keys.begin(), keys.end());
	}
	index.contig_offset[n_contigs] = index.keys.size();
	// bucket(key) = features with start <= key <= end; count, then fill in ascending feature id
	std::vector<uint32_t> count(index.keys.size() + 1, 0);
	for (int pass = 0; pass < 2; ++pass) {
		for (size_t f = 0; f < features.size(); ++f) {
			contig_t contig = features[f].contig;
			if (contig >= n_contigs) continue;
			const position_t* begin = &index.keys[0] + index.contig_offset[contig];
			const position_t* end = &index.keys[0] + index.contig_offset[contig + 1];
			size_t first = std::lower_bound(begin, end, features[f].start) - &index.keys[0];
			size_t last = std::lower_bound(begin, end, features[f].end) - &index.keys[0]; // exists: feature.end is a key
			for (size_t k = first; k <= last; ++k) {
				if (pass == 0) count[k]++;
				else index.members[index.member_offset[k] + count[k]++] = f;
			}
		}
		if (pass == 0) {
			index.member_offset.assign(index.keys.size() + 1, 0);
			for (size_t k = 0; k < index.keys.size(); ++k)
				index.member_offset[k + 1] = index.member_offset[k] + count[k];
			index.members.assign(index.member_offset[index.keys.size()], 0);
			std::fill(count.begin(), count.end(), 0);
		}
	}
}
template void make_flat_index<GeneRecord>(const std::vector<GeneRecord>&, size_t, FlatIndex&);
template void make_flat_index<ExonRecord>(const std::vector<ExonRecord>&, size_t, FlatIndex&);
template void make_flat_index<ProteinDomain>(const std::vector<ProteinDomain>&, size_t, FlatIndex&);

uint32_t FlatIndex::lower_bound(contig_t contig, position_t position) const {
	const position_t* begin = &keys[0] + contig_offset[contig];
	const position_t* end = &keys[0] + contig_offset[contig + 1];
	return std::lower_bound(begin, end, position) - &keys[0];
}

// reference: source/arriba.cpp:166-184
void compute_exonic_length(Annotation& annotation, const FlatIndex& exon_index) {
	for (size_t contig = 0; contig < exon_index.n_contigs(); ++contig) {
		position_t region_start = 0;
		for (uint32_t k = exon_index.contig_begin(contig); k < exon_index.contig_end(contig); ++k) {
			int previous_gene = -1;
			for (uint32_t m = exon_index.member_offset[k]; m < exon_index.member_offset[k + 1]; ++m) {
				int gene = annotation.exons[exon_index.members[m]].gene;
				if (previous_gene != gene) {
					annotation.genes[gene].exonic_length += exon_index.keys[k] - region_start;
					previous_gene = gene;
				}
			}
			region_start = exon_index.keys[k];
		}
	}
	for (size_t g = 0; g < annotation.genes.size(); ++g)
		if (annotation.genes[g].exonic_length == 0)
			annotation.genes[g].exonic_length = annotation.genes[g].end - annotation.genes[g].start;
}

namespace {
void insert_sorted_unique(std::vector<uint32_t>& set, const uint32_t* begin, const uint32_t* end) {
	for (const uint32_t* p = begin; p != end; ++p) {
		std::vector<uint32_t>::iterator at = std::lower_bound(set.begin(), set.end(), *p);
		if (at == set.end() || *at != *p)
			set.insert(at, *p);
	}
}
}

// reference: source/annotation.t.hpp:47-101
void get_annotation_by_coordinate(contig_t contig, position_t start, position_t end, std::vector<uint32_t>& result, const FlatIndex& index) {
	result.clear();
	if ((size_t) contig >= index.n_contigs())
		return;
	uint32_t contig_begin = index.contig_begin(contig), contig_end = index.contig_end(contig);
	const uint32_t* members = index.members.empty() ? NULL : &index.members[0];
	if (start == end) {
		uint32_t k = index.lower_bound(contig, start);
		if (k != contig_end)
			result.assign(members + index.member_offset[k], members + index.member_offset[k + 1]);
		return;
	}
	if (start > end)
		std::swap(start, end);
	std::vector<uint32_t> result_start, result_end;
	uint32_t k = index.lower_bound(contig, start);
	if (k != contig_end) {
		result_start.assign(members + index.member_offset[k], members + index.member_offset[k + 1]);
		if (index.keys[k] - start <= 2) {
			++k;
			if (k != contig_end)
				insert_sorted_unique(result_start, members + index.member_offset[k], members + index.member_offset[k + 1]);
		}
	}
	k = index.lower_bound(contig, end);
	if (k != contig_end)
		result_end.assign(members + index.member_offset[k], members + index.member_offset[k + 1]);
	if (k != contig_begin && contig_end > contig_begin) {
		--k;
		if (end - index.keys[k] <= 2)
			insert_sorted_unique(result_end, members + index.member_offset[k], members + index.member_offset[k + 1]);
	}
	std::set_intersection(result_start.begin(), result_start.end(), result_end.begin(), result_end.end(), std::back_inserter(result));
	if (result.empty())
		std::set_union(result_start.begin(), result_start.end(), result_end.begin(), result_end.end(), std::back_inserter(result));
}

namespace {
// reference: source/annotation.cpp:379-401
bool bucket_has_exon_near_splice_site(int gene, bool upstream, position_t breakpoint, const Annotation& annotation, const FlatIndex& exon_index, uint32_t k) {
	for (uint32_t m = exon_index.member_offset[k]; m < exon_index.member_offset[k + 1]; ++m) {
		const ExonRecord& exon = annotation.exons[exon_index.members[m]];
		if (exon.gene != gene)
			continue;
		if (upstream && abs(exon.start - breakpoint) <= MAX_SPLICE_SITE_DISTANCE &&
		    (exon.previous_exon != -1 || exon.previous_exon == -1 && exon.next_exon == -1 && exon.coding_region_start != -1 || exon.start == exon.coding_region_start) ||
		    !upstream && abs(exon.end - breakpoint) <= MAX_SPLICE_SITE_DISTANCE &&
		    (exon.next_exon != -1 || exon.previous_exon == -1 && exon.next_exon == -1 && exon.coding_region_start != -1 || exon.end == exon.coding_region_end))
			return true;
	}
	return false;
}
}

// reference: source/annotation.cpp:404-429
bool is_breakpoint_spliced(int gene, bool upstream, position_t breakpoint, const Annotation& annotation, const FlatIndex& exon_index) {
	contig_t contig = annotation.genes[gene].contig;
	if ((size_t) contig >= exon_index.n_contigs() || exon_index.contig_begin(contig) == exon_index.contig_end(contig))
		return false;
	uint32_t at = exon_index.lower_bound(contig, breakpoint);
	if (at != exon_index.contig_end(contig)) {
		if (bucket_has_exon_near_splice_site(gene, upstream, breakpoint, annotation, exon_index, at))
			return true;
		if (at + 1 != exon_index.contig_end(contig) && bucket_has_exon_near_splice_site(gene, upstream, breakpoint, annotation, exon_index, at + 1))
			return true;
	}
	if (at != exon_index.contig_begin(contig) && bucket_has_exon_near_splice_site(gene, upstream, breakpoint, annotation, exon_index, at - 1))
		return true;
	return false;
}

// reference: source/annotation.cpp:570-618
int get_spliced_distance(contig_t contig, position_t position1, position_t position2, int gene, const Annotation& annotation, const FlatIndex& exon_index) {
	if (position1 > position2)
		std::swap(position1, position2);
	if ((size_t) contig >= exon_index.n_contigs() || exon_index.contig_begin(contig) == exon_index.contig_end(contig))
		return position2 - position1;
	uint32_t k = exon_index.lower_bound(contig, position1);
	const uint32_t contig_end = exon_index.contig_end(contig);
	int distance = 0;
	if (k != contig_end && exon_index.keys[k] < position2) {
		distance += exon_index.keys[k] - position1;
		position1 = exon_index.keys[k];
	}
	for (; k != contig_end && exon_index.keys[k] < position2; ++k) {
		if (exon_index.keys[k] < position1)
			continue;
		position_t best_start = -1, best_end = -1, best_skip = -1;
		for (uint32_t m = exon_index.member_offset[k]; m < exon_index.member_offset[k + 1]; ++m) {
			const ExonRecord& exon = annotation.exons[exon_index.members[m]];
			if (exon.gene != gene || exon.next_exon == -1 || annotation.exons[exon.next_exon].start > position2)
				continue;
			position_t exon_start = std::max(position1, exon.start);
			position_t exon_end = std::min(position2, exon.end);
			position_t exon_skip = annotation.exons[exon.next_exon].start - exon_start + 1;
			if (best_start == -1 || 1.0 * (exon_end - exon_start) / exon_skip < 1.0 * (best_end - best_start) / best_skip) {
				best_start = exon_start; best_end = exon_end; best_skip = exon_skip;
			}
		}
		if (best_start != -1) {
			distance += best_end - best_start + 1;
			position1 = best_start + best_skip - 1;
		}
	}
	distance += position2 - position1;
	return distance;
}

// ---- coverage -----------------------------------------------------------------------------------

// reference: source/read_stats.cpp:146-158
void Coverage::resize(const Contigs& contigs, const Assembly& assembly) {
	coverage.resize(contigs.size()); fragment_starts.resize(contigs.size()); fragment_ends.resize(contigs.size());
	for (size_t contig = 0; contig < assembly.sequence.size() && contig < contigs.size(); ++contig)
		if (!assembly.sequence[contig].empty()) {
			size_t windows = assembly.sequence[contig].size() / COVERAGE_RESOLUTION + 2;
			coverage[contig].resize(windows); fragment_starts[contig].resize(windows); fragment_ends[contig].resize(windows);
		}
}
// reference: source/read_stats.cpp:269-306
bool Coverage::fragment_starts_here(contig_t contig, position_t start, position_t end) const {
	if ((size_t) contig >= fragment_starts.size()) return false;
	for (int window = start / COVERAGE_RESOLUTION + 1; window <= end / COVERAGE_RESOLUTION; ++window) {
		if ((size_t) window >= fragment_starts[contig].size()) return false;
		if (fragment_starts[contig][window]) return true;
	}
	return false;
}
bool Coverage::fragment_ends_here(contig_t contig, position_t start, position_t end) const {
	if ((size_t) contig >= fragment_ends.size()) return false;
	for (int window = start / COVERAGE_RESOLUTION; window < end / COVERAGE_RESOLUTION; ++window) {
		if ((size_t) window >= fragment_ends[contig].size()) return false;
		if (fragment_ends[contig][window]) return true;
	}
	return false;
}
int Coverage::get_coverage(contig_t contig, position_t position, bool upstream) const {
	if ((size_t) contig >= coverage.size() || coverage[contig].empty()) return -1;
	if (upstream)
		return (position < COVERAGE_RESOLUTION) ? 0 : coverage[contig][position / COVERAGE_RESOLUTION - 1];
	return coverage[contig][position / COVERAGE_RESOLUTION + 1];
}

std::string Batch::sequence(unsigned slot, size_t i) const {
	std::string result;
	sequence_into(slot, i, result);
	return result;
}
void Batch::sequence_into(unsigned slot, size_t i, std::string& out) const {
	static const char codes[] = "=ACMGRSVTWYHKDBN";
	out.resize(seq_length[slot][i]);
	const uint8_t* packed = &seq_pool[0] + (size_t) seq_offset[slot][i] * 4;
	for (size_t b = 0; b < out.size(); ++b)
		out[b] = codes[(packed[b >> 1] >> ((~b & 1) << 2)) & 15];
}

// ---- blacklist / known-fusions files (reference: source/filter_blacklisted_ranges.cpp:17-118, the line loops at :244-264 and
// source/recover_known_fusions.cpp:21-37) --------------------------------------------------------

namespace {

// one field of the reference's tsv_stream_t (source/read_compressed_file.cpp:62-88): fails when the cursor is at or behind the end
struct FieldCursor {
	const std::string& data; char delimiter; size_t position; bool failed;
	FieldCursor(const std::string& text, char d): data(text), delimiter(d), position(0), failed(false) {}
	void next(std::string& out) {
		if (position >= data.size()) { failed = true; return; }
		const size_t start = position;
		position = data.find(delimiter, start);
		out = data.substr(start, position - start);
		if (position < data.size()) position++;
	}
	void next(int& out) {
		std::string field;
		const bool was_failed = failed;
		next(field);
		if (failed && !was_failed) return;
		// reference: str_to_int (source/common.hpp:316-321)
		const char* text = field.c_str();
		char* end_of_parsing;
		const long value = strtol(text, &end_of_parsing, 10);
		out = (int) value;
		if (!(*text != ' ' && end_of_parsing != text && *end_of_parsing == '\0' && value != LONG_MAX && value != LONG_MIN)) failed = true;
	}
};

void warn_malformed_range(const std::string& range) { fprintf(stderr, "WARNING: unknown gene or malformed range: %s\n", range.c_str()); }

// reference: parse_range (:17-80)
bool parse_range(std::string range, const Contigs& contigs, agpu_range_item& item) {
	const std::string original = range;
	const size_t separator = range.find_last_of(':'); // the last colon: contig names may hold colons
	if (separator < range.size()) range[separator] = '\t';
	FieldCursor fields(range, '\t');
	std::string contig_name, start_and_end;
	fields.next(contig_name); fields.next(start_and_end);
	if (fields.failed || contig_name.empty() || start_and_end.empty()) { warn_malformed_range(range); return false; }
	item.strand_defined = contig_name[0] == '+' || contig_name[0] == '-';
	if (item.strand_defined) { item.strand = contig_name[0] == '+'; contig_name = contig_name.substr(1); }
	contig_name = remove_chr(contig_name);
	std::map<std::string, contig_t>::const_iterator contig;
	if (contig_name.size() >= 2 && contig_name[contig_name.size() - 1] == '*') { // a trailing asterisk: the closest match
		contig_name = contig_name.substr(0, contig_name.size() - 1);
		contig = contigs.by_name.lower_bound(contig_name);
		if (contig != contigs.by_name.end() && contig_name != contig->first.substr(0, contig_name.size())) contig = contigs.by_name.end();
	} else {
		contig = contigs.by_name.find(contig_name);
		if (contig == contigs.by_name.end()) warn_malformed_range(range);
	}
	if (contig == contigs.by_name.end()) return false;
	item.contig = contig->second;
	FieldCursor positions(start_and_end, '-');
	int start = 0, end = 0;
	if (start_and_end.find('-') < start_and_end.size()) { // contig:start-end
		positions.next(start); positions.next(end);
		if (positions.failed) { warn_malformed_range(range); return false; }
		item.start = start - 1; item.end = end - 1; // zero-based
	} else { // contig:position
		positions.next(start);
		if (positions.failed) { warn_malformed_range(range); return false; }
		item.start = item.end = start - 1;
	}
	return true;
}

// reference: parse_blacklist_item (:83-118)
bool parse_range_item(const std::string& text, agpu_range_item& item, const Contigs& contigs, const Annotation& annotation, bool allow_keyword) {
	memset(&item, 0, sizeof(item));
	if (text.empty()) { fprintf(stderr, "WARNING: encountered a line with an empty column => skipped\n"); return false; }
	if (allow_keyword) {
		static const struct { const char* word; uint8_t type; } keywords[] = { { "any", AGPU_RULE_ANY }, { "split_read_donor", AGPU_RULE_SPLIT_READ_DONOR }, { "split_read_acceptor", AGPU_RULE_SPLIT_READ_ACCEPTOR },
			{ "split_read_any", AGPU_RULE_SPLIT_READ_ANY }, { "discordant_mates", AGPU_RULE_DISCORDANT_MATES }, { "read_through", AGPU_RULE_READ_THROUGH }, { "low_support", AGPU_RULE_LOW_SUPPORT },
			{ "filter_spliced", AGPU_RULE_FILTER_SPLICED }, { "not_both_spliced", AGPU_RULE_NOT_BOTH_SPLICED } };
		for (size_t k = 0; k < sizeof(keywords) / sizeof(keywords[0]); ++k)
			if (text == keywords[k].word) { item.type = keywords[k].type; return true; }
	}
	std::unordered_map<std::string, int>::const_iterator gene = annotation.gene_by_name.find(text);
	if (gene != annotation.gene_by_name.end()) {
		const GeneRecord& record = annotation.genes[gene->second];
		item.type = AGPU_RULE_GENE; item.gene = (uint32_t) gene->second; item.contig = record.contig; item.start = record.start; item.end = record.end;
		return true;
	}
	if (!parse_range(text, contigs, item)) return false;
	item.type = item.start == item.end ? AGPU_RULE_POSITION : AGPU_RULE_RANGE;
	return true;
}

}

void load_range_rules(const std::string& path, const Contigs& contigs, const Annotation& annotation, bool allow_keyword_in_second_column, std::vector<agpu_range_rule>& rules) {
	rules.clear();
	LineReader file(path);
	std::string line;
	while (file.getline(line)) {
		if (line.empty() || line[0] == '#') continue;
		FieldCursor fields(line, '\t');
		std::string range1, range2;
		fields.next(range1); fields.next(range2);
		agpu_range_rule rule;
		if (!parse_range_item(range1, rule.first, contigs, annotation, false) || !parse_range_item(range2, rule.second, contigs, annotation, allow_keyword_in_second_column)) continue;
		rules.push_back(rule);
	}
}

// ---- tags and protein domains (columns of the output file) ----------------------------------------

namespace {
// reference: get_genome_bins_from_range (source/filter_blacklisted_ranges.cpp:221-225)
void add_to_genome_bins(const agpu_range_item& item, uint32_t rule, std::map<uint64_t, std::vector<uint32_t> >& by_bin) {
	const int bin_size = 100000;
	for (position_t bin = item.start / bin_size; bin <= (item.end + bin_size - 1) / bin_size; ++bin) by_bin[(uint64_t) item.contig << 32 | (uint32_t) (bin * bin_size)].push_back(rule);
}
// reference: get_gff3_attribute (source/annotate_protein_domains.cpp:14-31)
bool get_gff3_attribute(const std::string& attributes, const std::string& name, std::string& value) {
	size_t start = attributes.find(name + "=");
	if (start >= attributes.size()) { fprintf(stderr, "WARNING: failed to extract %s from line in GFF3 file: %s\n", name.c_str(), attributes.c_str()); return false; }
	start += name.size() + 1;
	size_t end = attributes.find(';', start);
	if (end >= attributes.size()) end = attributes.size();
	value = attributes.substr(start, end - start);
	return true;
}
bool is_hex_digit(char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }
}

void load_tags(const std::string& path, const Contigs& contigs, const Annotation& annotation, Tags& tags) {
	tags.rules.clear(); tags.by_bin.clear();
	LineReader file(path);
	std::string line;
	while (file.getline(line)) {
		if (line.empty() || line[0] == '#') continue;
		FieldCursor fields(line, '\t');
		std::string range1, range2, tag;
		fields.next(range1); fields.next(range2); fields.next(tag);
		if (tag.empty()) { fprintf(stderr, "WARNING: encountered a line with an empty tag => skipped\n"); continue; }
		TagRule rule;
		if (!parse_range_item(range1, rule.first, contigs, annotation, false) || !parse_range_item(range2, rule.second, contigs, annotation, false)) continue;
		for (size_t k = 0; k < tag.size(); ++k) if (tag[k] < '!' || tag[k] > '~' || tag[k] == ',') tag[k] = '_'; // characters with a meaning in the output format
		rule.tag = tag;
		tags.rules.push_back(rule);
		add_to_genome_bins(rule.first, (uint32_t) tags.rules.size() - 1, tags.by_bin);
		add_to_genome_bins(rule.second, (uint32_t) tags.rules.size() - 1, tags.by_bin);
	}
}

void load_protein_domains(const std::string& path, const Contigs& contigs, const Annotation& annotation, std::vector<ProteinDomain>& domains, FlatIndex& index) {
	domains.clear();
	std::unordered_map<std::string, int> gene_by_id; // later genes win (source/annotate_protein_domains.cpp:36-38)
	for (size_t g = 0; g < annotation.genes.size(); ++g) gene_by_id[strip_ensembl_version_number(annotation.genes[g].gene_id)] = (int) g;
	LineReader file(path);
	std::string line;
	std::set<std::string> unknown_genes;
	while (file.getline(line)) {
		if (line.empty() || line[0] == '#') continue;
		FieldCursor fields(line, '\t');
		ProteinDomain domain;
		std::string contig, strand, attributes, gene_name, gene_id, trash;
		int start = 0, end = 0;
		fields.next(contig); fields.next(trash); fields.next(trash); fields.next(start); fields.next(end); fields.next(trash); fields.next(strand); fields.next(trash); fields.next(attributes);
		if (fields.failed || contig.empty() || strand.empty() || attributes.empty()) { fprintf(stderr, "WARNING: failed to parse line in GFF3 file: %s\n", line.c_str()); continue; }
		if (!get_gff3_attribute(attributes, "gene_name", gene_name) || !get_gff3_attribute(attributes, "gene_id", gene_id) || !get_gff3_attribute(attributes, "Name", domain.name)) continue;
		std::map<std::string, contig_t>::const_iterator known_contig = contigs.by_name.find(remove_chr(contig));
		if (known_contig == contigs.by_name.end()) { fprintf(stderr, "WARNING: unknown contig: %s\n", contig.c_str()); continue; }
		// "%2C" and the like stand for special characters
		for (std::string::size_type pos = domain.name.find("%"); pos < domain.name.size(); pos = domain.name.find("%", pos + 1))
			if (pos + 2 < domain.name.size() && is_hex_digit(domain.name[pos + 1]) && is_hex_digit(domain.name[pos + 2]))
				domain.name = domain.name.substr(0, pos) + (char) strtoul(domain.name.substr(pos + 1, 2).c_str(), NULL, 16) + domain.name.substr(pos + 3);
		for (size_t k = 0; k < domain.name.size(); ++k) if (domain.name[k] < '!' || domain.name[k] > '~' || domain.name[k] == ',' || domain.name[k] == '|') domain.name[k] = '_';
		std::unordered_map<std::string, int>::const_iterator by_id = gene_by_id.find(strip_ensembl_version_number(gene_id));
		if (by_id != gene_by_id.end()) domain.gene = by_id->second;
		else {
			std::unordered_map<std::string, int>::const_iterator by_name = annotation.gene_by_name.find(gene_name);
			if (by_name == annotation.gene_by_name.end()) {
				if (unknown_genes.insert(gene_name + " " + gene_id).second) fprintf(stderr, "WARNING: unknown gene: %s %s\n", gene_name.c_str(), gene_id.c_str()); // once per gene
				continue;
			}
			domain.gene = by_name->second;
		}
		domain.contig = known_contig->second; domain.start = start - 1; domain.end = end - 1; domain.strand = strand[0] == '+';
		domains.push_back(domain);
	}
	if (domains.empty()) throw std::runtime_error("failed to parse GFF3 file");
	make_flat_index(domains, std::max(domains.size(), contigs.size()), index);
}

// ---- structural variants from whole-genome sequencing (-d; reference: source/filter_genomic_support.cpp:15-165) -----------------------

namespace {
// reference: parse_breakpoint (:15-36)
bool parse_genomic_breakpoint(std::string text, const Contigs& contigs, uint32_t& contig, int32_t& position) {
	const size_t separator = text.find_last_of(':');
	if (separator < text.size()) text[separator] = '\t';
	FieldCursor fields(text, '\t');
	std::string contig_name;
	fields.next(contig_name);
	std::map<std::string, contig_t>::const_iterator known = contigs.by_name.find(remove_chr(contig_name));
	if (known == contigs.by_name.end()) return false;
	contig = known->second;
	int value = 0;
	fields.next(value);
	if (fields.failed) return false;
	position = value - 1;
	return true;
}
bool parse_direction(const std::string& text, bool& upstream) {
	if (text == "upstream" || text == "-") upstream = true; else if (text == "downstream" || text == "+") upstream = false; else return false;
	return true;
}
// reference: parse_vcf_info (:49-60)
bool parse_vcf_info(const std::string& info, const std::string& field, std::string& value) {
	size_t start;
	if (info.substr(0, field.size() + 1) == field + "=") start = field.size() + 1;
	else { start = info.find(";" + field + "="); if (start >= info.size()) return false; start += field.size() + 2; }
	value = info.substr(start, info.find(';', start) - start);
	return true;
}
}

void load_genomic_breakpoints(const std::string& path, const Contigs& contigs, std::vector<agpu_genomic_breakpoint>& variants) {
	variants.clear();
	LineReader file(path);
	std::string line;
	while (file.getline(line)) {
		if (line.empty() || line[0] == '#') continue;
		// the four-column format first: contig1:position1, contig2:position2, direction1, direction2
		FieldCursor fields(line, '\t');
		std::string breakpoint1, breakpoint2, direction_text1, direction_text2, sv_type;
		fields.next(breakpoint1); fields.next(breakpoint2); fields.next(direction_text1); fields.next(direction_text2);
		uint32_t contig1 = 0, contig2 = 0; int32_t position1 = 0, position2 = 0; bool upstream1 = false, upstream2 = false;
		bool parsed = parse_genomic_breakpoint(breakpoint1, contigs, contig1, position1) && parse_genomic_breakpoint(breakpoint2, contigs, contig2, position2) && parse_direction(direction_text1, upstream1) && parse_direction(direction_text2, upstream2);
		if (!parsed) { // then VCF
			FieldCursor vcf(line, '\t');
			std::string chrom, pos, alt, info, filter, ignore;
			vcf.next(chrom); vcf.next(pos); vcf.next(ignore); vcf.next(ignore); vcf.next(alt); vcf.next(ignore); vcf.next(filter); vcf.next(info);
			bool failed = !parse_vcf_info(info, "SVTYPE", sv_type), skip = false;
			if (!failed && sv_type == "BND") {
				const size_t opening = alt.find('['), closing = alt.find(']');
				const char bracket = opening < closing ? '[' : ']';
				const size_t first = std::min(opening, closing), second = alt.find(bracket, first + 1);
				if (first >= alt.size() || second >= alt.size()) {
					if (!alt.empty() && (alt[0] == '.' || alt[alt.size() - 1] == '.')) skip = true; // a single breakend: silently ignored
					else failed = true;
				} else {
					upstream1 = first == 0; upstream2 = bracket == '[';
					breakpoint2 = alt.substr(first + 1, second - first - 1);
				}
			} else if (!failed) {
				std::string end;
				if (!parse_vcf_info(info, "END", end)) failed = true;
				else {
					breakpoint2 = chrom + ":" + end;
					if (sv_type == "INV") { upstream1 = false; upstream2 = false; }
					else if (sv_type == "DEL") { upstream1 = false; upstream2 = true; }
					else if (sv_type == "DUP") { upstream1 = true; upstream2 = false; }
					else failed = true;
				}
			}
			if (skip) continue;
			if (!failed && (!parse_genomic_breakpoint(chrom + ":" + pos, contigs, contig1, position1) || !parse_genomic_breakpoint(breakpoint2, contigs, contig2, position2))) failed = true;
			if (failed) { fprintf(stderr, "WARNING: failed to parse line: %s\n", line.c_str()); continue; }
			if (filter != "PASS") continue;
		}
		if (contig2 < contig1 || (contig2 == contig1 && position2 < position1)) { std::swap(contig1, contig2); std::swap(position1, position2); std::swap(upstream1, upstream2); } // indexed by the smaller coordinate
		agpu_genomic_breakpoint variant = { contig1, contig2, position1, position2, (uint8_t) upstream1, (uint8_t) upstream2, { 0, 0 } };
		variants.push_back(variant);
		if (sv_type == "INV") { variant.upstream1 = variant.upstream2 = 1; variants.push_back(variant); } // the VCF type INV stands for two breakpoints
	}
}

}
