// arriba_amd/csrc/workflow/main.cpp -- the command line of the reference (source/options.cpp:270-481: flags, defaults, checks and messages) in front of
// arriba_workflow_run, so that run_arriba.sh can call this binary where it calls `arriba` (run_arriba.sh:42: STAR ... | arriba -x /dev/stdin -o ... -O ...).
// Exit code 0 on success, 1 on any error ("ERROR: ..." on stderr, as crash() does in the reference).  Not supported and said so: -c (separate chimeric
// SAM file of old STAR versions), SAM text and CRAM input, -@ (the threads of the BAM decompression are chosen by the ingest).
// The form `arriba_gpu_workflow assembly.fa annotation.gtf chimeric.bam fusions.tsv [discarded.tsv [...]]` of round 1 (no dash in front of the first
// argument) is still understood.
#include "../../../include/arriba_workflow.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <libgen.h>
#include <sstream>
#include <string>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

const char* const FILTER_NAMES[AGPU_FILTER_COUNT] = { "", "duplicates", "inconsistently_clipped", "homopolymer", "read_through", "same_gene", "small_insert_size", "long_gap", "hairpin", "multimappers", "mismatches", "mismappers", "relative_support",
	"intronic", "non_coding_neighbors", "intragenic_exonic", "internal_tandem_duplication", "min_support", "known_fusions", "spliced", "blacklist", "end_to_end", "in_vitro", "merge_adjacent", "select_best", "marginal_read_through", "short_anchor",
	"no_coverage", "many_spliced", "no_genomic_support", "uninteresting_contigs", "viral_contigs", "top_expressed_viral_contigs", "low_coverage_viral_contigs", "genomic_support", "isoforms", "low_entropy", "homologs" }; // source/common.hpp:29-67

[[noreturn]] void crash(const std::string& message) { std::cerr << "ERROR: " << message << std::endl; exit(1); }
void require(bool condition, const std::string& message) { if (!condition) crash(message); }

bool parse_int(const char* text, long& value) { char* end = NULL; errno = 0; value = strtol(text, &end, 10); return *text != '\0' && *end == '\0' && errno == 0; }
bool parse_float(const char* text, float& value) { char* end = NULL; errno = 0; value = strtof(text, &end); return *text != '\0' && *end == '\0' && errno == 0; }
void int_option(char flag, const char* text, uint32_t& target, long minimum, long maximum, const std::string& complaint) {
	long value;
	require(parse_int(text, value) && value >= minimum && value <= maximum, std::string("argument to -") + flag + " " + complaint);
	target = (uint32_t) value;
}
void float_option(char flag, const char* text, float& target, float minimum, float maximum, const std::string& complaint) {
	require(parse_float(text, target) && target >= minimum && target <= maximum, std::string("argument to -") + flag + " " + complaint);
}
void readable(const char* path) { require(strcmp(path, "-") == 0 || access(path, R_OK) == 0, std::string("file not found/readable: ") + path); }
void parent_exists(const char* path) {
	std::string copy(path);
	struct stat status;
	require(stat(dirname(&copy[0]), &status) == 0 && S_ISDIR(status.st_mode), std::string("parent directory of output file '") + path + "' does not exist");
}

void usage() {
	std::cout << "arriba_gpu_workflow -- MI355X-native fusion caller with the command line and the output files of Arriba 2.5.1\n\n"
	             "Usage: arriba_gpu_workflow -x Aligned.out.bam -g annotation.gtf -a assembly.fa -o fusions.tsv [-b blacklists.tsv] [-k known_fusions.tsv] [-t tags.tsv] [-p protein_domains.gff3]\n"
	             "                           [-d structural_variants.tsv] [-O fusions.discarded.tsv] [OPTIONS]\n\n"
	             " -x FILE  BAM file with the main alignments of STAR run with --chimOutType WithinBAM (BGZF, gzip or uncompressed; a file, a pipe, /dev/stdin or -)\n"
	             " -g FILE  gene annotation (GTF)        -a FILE  assembly (FastA)          -o FILE  output file             -O FILE  discarded fusions\n"
	             " -b FILE  blacklist                    -k FILE  known fusions             -t FILE  tags                    -p FILE  protein domains (GFF3)\n"
	             " -d FILE  structural variants from WGS -D INT   max. distance to them (100000)\n"
	             " -s {yes,no,auto,reverse} strandedness (auto)   -i CONTIGS interesting contigs   -v CONTIGS viral contigs   -G FEATURES GTF features\n"
	             " -f FILTERS filters to switch off, comma/space separated:";
	for (int f = 1; f < AGPU_FILTER_COUNT; ++f) std::cout << " " << FILTER_NAMES[f];
	std::cout << "\n -E FLOAT e-value cutoff (0.3)   -S INT min. support (2)   -m FLOAT max. mis-mapper fraction (0.8)   -L FLOAT max. homolog identity (0.3)   -H INT homopolymer length (6)\n"
	             " -R INT min. read-through distance (10000)   -A INT min. anchor length (23)   -M INT min. spliced events (4)   -K FLOAT max. k-mer content (0.6)   -V FLOAT mismatch p-value (0.01)\n"
	             " -F INT fragment length for single-end data (200)   -U INT subsampling threshold (300)   -Q FLOAT high-expression quantile (0.998)   -e FLOAT exonic fraction (0.33)\n"
	             " -T INT top viral contigs (5)   -C FLOAT min. covered fraction of a viral contig (0.05)   -l INT max. ITD length (100)   -z FLOAT min. ITD allele fraction (0.07)   -Z INT min. ITD support (10)\n"
	             " -u duplicates are marked in the BAM file   -X extra columns for discarded fusions   -I fill gaps of the fusion transcript from the assembly   -h this text\n"
	             " --device N  the GPU to use (0)            --host-ingest  read_chimeric_alignments on the host instead of on the GPU\n";
}

}

int main(int argc, char** argv) {
	time_t start_time;
	time(&start_time);
	arriba_workflow_options options;
	arriba_workflow_default_options(&options);
	if (argc >= 5 && argv[1][0] != '-') { // round 1's positional form
		const char** slots[] = { &options.assembly_file, &options.gene_annotation_file, &options.chimeric_bam_file, &options.output_file, &options.discarded_output_file, &options.blacklist_file, &options.known_fusions_file,
		                         &options.tags_file, &options.protein_domains_file, &options.genomic_breakpoints_file };
		for (int a = 1; a < argc && a <= 10; ++a) *slots[a - 1] = strcmp(argv[a], "-") == 0 ? NULL : argv[a];
		arriba_workflow_report report;
		if (arriba_workflow_run(&options, &report) != 0) { fprintf(stderr, "%s\n", arriba_workflow_last_error()); return 1; }
		for (uint32_t s = 0; s < report.n_stages; ++s) printf("%s\t%llu\n", report.stages[s].stage, (unsigned long long) report.stages[s].count);
		return 0;
	}
	std::cout << "[" << "arriba_gpu_workflow" << "] MI355X-native fusion caller (command line, progress lines and output files of Arriba 2.5.1)" << std::endl;
	if (argc == 1) { usage(); crash("no arguments given"); }
	require(argv[1][0] == '-' && argv[1][1] != '\0', std::string("cannot interpret the first argument: ") + argv[1]);
	// the two long options of this implementation are taken out first; the rest is the reference's getopt string (source/options.cpp:282)
	std::vector<char*> arguments(1, argv[0]);
	for (int a = 1; a < argc; ++a) {
		if (strcmp(argv[a], "--host-ingest") == 0) options.host_ingest = 1;
		else if (strcmp(argv[a], "--device") == 0 && a + 1 < argc) { long device; require(parse_int(argv[++a], device) && device >= 0, "invalid argument to --device"); options.device_index = (int) device; }
		else arguments.push_back(argv[a]);
	}
	std::string interesting_contigs, viral_contigs;
	bool seen[256]; memset(seen, 0, sizeof(seen));
	bool blacklist_enabled = true;
	opterr = 0;
	const std::string valid_arguments = "c:x:d:g:G:o:O:t:p:a:b:k:s:i:v:f:E:S:m:L:H:D:R:A:M:K:V:F:U:Q:e:T:C:l:z:Z:@:uXIh";
	int c;
	const int n_arguments = (int) arguments.size();
	while ((c = getopt(n_arguments, arguments.data(), valid_arguments.c_str())) != -1) {
		if (c != '?') { require(!seen[(unsigned char) c], std::string("option -") + (char) c + " specified too often"); seen[(unsigned char) c] = true; }
		uint32_t number = 0;
		switch (c) {
			case 'c': crash("option -c (chimeric alignments in a separate SAM file, STAR < 2.5.3a) is not supported: run STAR with --chimOutType WithinBAM and pass the main BAM file with -x");
			case 'x': {
				const std::string path(optarg);
				require(!(path.size() >= 5 && path.substr(path.size() - 5) == ".cram") && !(path.size() >= 4 && path.substr(path.size() - 4) == ".sam"), "only BAM input is supported (SAM text and CRAM are not): " + path);
				options.chimeric_bam_file = optarg; readable(optarg); break;
			}
			case 'd': options.genomic_breakpoints_file = optarg; readable(optarg); break;
			case 'g': options.gene_annotation_file = optarg; readable(optarg); break;
			case 'G': options.gtf_features = optarg; break; // (checked when the annotation is loaded)
			case 'o': options.output_file = optarg; parent_exists(optarg); break;
			case 'O': options.discarded_output_file = optarg; parent_exists(optarg); break;
			case 't': options.tags_file = optarg; readable(optarg); break;
			case 'p': options.protein_domains_file = optarg; readable(optarg); break;
			case 'a': options.assembly_file = optarg; readable(optarg); break;
			case 'b': options.blacklist_file = optarg; readable(optarg); break;
			case 'k': options.known_fusions_file = optarg; readable(optarg); break;
			case 's':
				if (!strcmp(optarg, "auto")) options.device.strandedness = 3; else if (!strcmp(optarg, "yes")) options.device.strandedness = 1; else if (!strcmp(optarg, "no")) options.device.strandedness = 0;
				else if (!strcmp(optarg, "reverse")) options.device.strandedness = 2; else crash(std::string("invalid type of strandedness: ") + optarg);
				break;
			case 'i': interesting_contigs = optarg; for (size_t k = 0; k < interesting_contigs.size(); ++k) if (interesting_contigs[k] == ',') interesting_contigs[k] = ' '; options.interesting_contigs = interesting_contigs.c_str(); break;
			case 'v': viral_contigs = optarg; for (size_t k = 0; k < viral_contigs.size(); ++k) if (viral_contigs[k] == ',') viral_contigs[k] = ' '; options.viral_contigs = viral_contigs.c_str(); break;
			case 'f': {
				std::string list(optarg);
				for (size_t k = 0; k < list.size(); ++k) if (list[k] == ',') list[k] = ' ';
				std::istringstream words(list);
				std::string word;
				while (words >> word) {
					int id = 0;
					for (int f = 1; f < AGPU_FILTER_COUNT; ++f) if (word == FILTER_NAMES[f]) id = f;
					require(id != 0, "invalid argument to option -f: " + word);
					options.device.filter_enabled[id] = 0;
					if (word == "blacklist") blacklist_enabled = false;
				}
				break;
			}
			case 'E': float_option('E', optarg, options.device.evalue_cutoff, 0, 1e38f, "must be greater than 0"); break;
			case 'S': int_option('S', optarg, options.device.min_support, 0, INT_MAX, "is invalid"); break;
			case 'm': float_option('m', optarg, options.device.max_mismapper_fraction, 0, 1, "must be between 0 and 1"); break;
			case 'L': float_option('L', optarg, options.max_homolog_identity, 0, 1, "must be between 0 and 1"); break;
			case 'H': int_option('H', optarg, options.device.homopolymer_length, 2, INT_MAX, "must be greater than 1"); break;
			case 'D': int_option('D', optarg, number, 0, INT_MAX, "is invalid"); options.max_genomic_breakpoint_distance = (int32_t) number; break;
			case 'R': int_option('R', optarg, options.device.min_read_through_distance, 0, INT_MAX, "is invalid"); break;
			case 'A': int_option('A', optarg, options.min_anchor_length, 0, INT_MAX, "is invalid"); break;
			case 'M': int_option('M', optarg, options.min_spliced_events, 0, INT_MAX, "is invalid"); break;
			case 'K': float_option('K', optarg, options.device.max_kmer_content, 0, 1, "must be between 0 and 1"); break;
			case 'V': float_option('V', optarg, options.device.mismatch_pvalue_cutoff, 0, 1, "must be between 0 and 1"); break;
			case 'F': int_option('F', optarg, options.device.fragment_length, 1, INT_MAX, "must be an integer greater than 0"); break;
			case 'U': int_option('U', optarg, options.device.subsampling_threshold, 1, SHRT_MAX, "must be an integer between 1 and " + std::to_string(SHRT_MAX)); break;
			case 'Q': float_option('Q', optarg, options.high_expression_quantile, 0, 1, "must be between 0 and 1"); break;
			case 'e': float_option('e', optarg, options.device.exonic_fraction, 0, 1, "must be between 0 and 1"); break;
			case 'T': int_option('T', optarg, options.top_viral_contigs, 1, INT_MAX, "is invalid"); break;
			case 'C': float_option('C', optarg, options.viral_contig_min_covered_fraction, 0, 1, "must be between 0 and 1"); break;
			case 'l': int_option('l', optarg, options.device.max_itd_length, 1, INT_MAX, "must be an integer greater than 0"); break;
			case 'z': float_option('z', optarg, options.min_itd_allele_fraction, 0, 1, "must be between 0 and 1"); break;
			case 'Z': int_option('Z', optarg, options.min_itd_support, 1, INT_MAX, "must be an integer greater than 0"); break;
			case '@': int_option('@', optarg, number, 1, INT_MAX, "must be an integer greater than 0"); break; // accepted; the ingest picks its own threads
			case 'u': options.device.external_duplicate_marking = 1; break;
			case 'X': options.print_extra_info_for_discarded_fusions = 1; break;
			case 'I': options.fill_sequence_gaps = 1; break;
			case 'h': usage(); return 0;
			default:
				require(valid_arguments.find(std::string(1, (char) optopt) + ":") == std::string::npos, std::string("option -") + (char) optopt + " requires an argument");
				crash(std::string("unknown option: -") + (char) optopt);
		}
		require(!(optind < n_arguments && (arguments[optind][0] == '\0' || arguments[optind][0] != '-')), std::string("option -") + (char) c + " has too many arguments (arguments with blanks must be wrapped in quotes)");
	}
	require(options.chimeric_bam_file != NULL, "missing mandatory option -x");
	require(options.gene_annotation_file != NULL, "missing mandatory option -g");
	require(options.output_file != NULL, "missing mandatory option -o");
	require(options.assembly_file != NULL, "missing mandatory option -a");
	require(!(blacklist_enabled && options.blacklist_file == NULL), "filter 'blacklist' enabled, but missing option -b (use '-f blacklist' if you want to disable the blacklist)");
	if (!options.device.filter_enabled[30]) options.interesting_contigs = "*"; // all contigs are loaded when the filter is off, whatever -i says (source/arriba.cpp:92-93)
	options.log_to_stdout = 1;
	if (arriba_workflow_run(&options, NULL) != 0) { std::cerr << arriba_workflow_last_error() << std::endl; return 1; }
	// source/arriba.cpp:612-628
	time_t end_time;
	time(&end_time);
	struct rusage usage_now;
	getrusage(RUSAGE_SELF, &usage_now);
	auto hhmmss = [](unsigned long long seconds) { std::ostringstream text; text << std::setfill('0') << std::setw(2) << (seconds / 3600) << ":" << std::setw(2) << (seconds % 3600 / 60) << ":" << std::setw(2) << (seconds % 60); return text.str(); };
	char stamp[100];
	strftime(stamp, sizeof(stamp), "[%Y-%m-%dT%X]", localtime(&end_time));
	std::cout << stamp << " Done (elapsed time=" << hhmmss((unsigned long long) difftime(end_time, start_time)) << ", CPU time=" << hhmmss(usage_now.ru_utime.tv_sec + usage_now.ru_stime.tv_sec) << ", peak memory=" << std::setprecision(3)
	          << (usage_now.ru_maxrss / (1024.0 * 1024)) << "gb)" << std::endl;
	return 0;
}
