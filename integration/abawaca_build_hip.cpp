// abawaca_build_hip — the reference-side binding of libabawaca_hip.so for `abawaca-build`.
//
// A replacement for the `main` of the reference's src/abawaca-build.cpp that keeps everything the reference does on the
// host with the reference's OWN classes -- the command line (ref :449-479), the FASTA reader
// (Bio::SeqIORead_fasta<Bio::DNASequence>, ref src/SeqIORead_fasta.h:51-103, with DNAString's upper-casing and its
// exception for a lower-case n, ref src/String.cpp:37-51), the SAM reader and record parser (Bio::SAMReader,
// Bio::ReadMapping: ref src/ReadMappingReader.cpp:80-116, src/ReadMapping.cpp:23-185; only as the fall-back for a file
// that abw_sam_parse refuses, or with ABW_SAM_READER=reference), the read filter (ref :546), the
// scaffold order of its std::map (ref :482,596), and the text of abawaca.names / abawaca.info / the abawaca.lrn header
// (ref :578-599) -- and hands the three data-parallel pieces to the GPU through the C ABI:
//
//     Scaf::Scaf + Scaf_segment::Scaf_segment   (ref :103-228)      ->  abw_sig_build       (one call for the assembly)
//     SAMReader + ReadMapping + the filter      (see above)          ->  abw_sam_parse       (per 64 MiB of a SAM file)
//     Scaf::add_mapped_read / Scaf_segment::... (ref :231-244,177-185) ->  abw_cov_accumulate  (per chunk of a SAM file)
//     the row loop of abawaca.lrn               (ref :600-605)      ->  abw_lrn_emit
//
// It is linked with the reference's unmodified ReadMapping.cpp, ReadMappingReader.cpp, common.cpp and String.cpp
// (oracle/Makefile: oracle/_ref/hip_build_driver).  tests/test_gpu_build_driver.py runs it beside the reference's own
// abawaca-build and requires byte-identical abawaca.lrn / .names / .info.
//
// What this file is: a PATCH of the reference's main, written out as a whole file because Scaf / Scaf_segment are file-local
// to src/abawaca-build.cpp and cannot be linked.  About 45 of its lines are therefore the reference's own main with small
// edits, kept because they ARE the command line and the wire formats: the option loop (ref :449-479), the four output names
// and the mkdir (ref :492-505), the SAM loop of the fall-back reader with its filter (ref :522-552, the links bookkeeping
// removed), the header fprintfs of abawaca.lrn and the per-segment / per-scaffold lines of abawaca.names / abawaca.info
// (ref :578-599), data.txt (ref :615-630).  Everything else -- the batching of the FASTA loop, the chunked SAM text, the
// bookkeeping between the three library calls -- is this repository's.  INTEGRATION.md section 2 shows the same change as a
// diff against src/abawaca-build.cpp.
//
// Not done here, as in the rest of this repository: the paired-end links file (ref :249-367,519-557) and the
// gene / SCG prediction that shells out to prodigal (ref :401-431); data.txt is written with the same keys.
//
// Two deliberate differences, both where the reference's behaviour is undefined:
//   * a segment that received no read in the LAST samples is printed with 0.000 in those columns; the reference reads
//     its vector out of bounds there (get_dimension, ref :188-193 against the resize of :179-180);
//   * two FASTA records with the same name: the reference counts the segments of both in the header but keeps only the
//     first in its map; here the second is refused.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <iostream>
#include <map>
#include <string>
#include <vector>
#include "ReadMappingReader.h"
#include "SeqIORead_fasta.h"
#include "String.h"
#include "common.h"
#include "abawaca_hip.h"
#include <stdlib.h>
// the GPU this process works on: ABW_DEVICE in the environment (default 0)
static inline int abw_env_device() { const char* e = getenv("ABW_DEVICE"); return e ? atoi(e) : 0; }

using namespace std;

namespace {

void die(abw_ctx* ctx, const char* what)
{
	cerr << "abawaca-build (hip): " << what << ": " << (ctx ? abw_last_error(ctx) : "no context") << endl;
	exit(-1);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ScafRec {
	string		name;
	size_t		len, ns;
	double		gc;
	size_t		nbps;		// Scaf::nbps, ref :242-243
};

}	// namespace

int main(int argc, const char* argv[])
{
	const size_t	window_size = 2000;			// ref :436
	const size_t	max_snps = 15;				// ref :437
	const size_t	sam_chunk = (size_t)1 << 22;		// mappings per abw_cov_accumulate call

	if(argc < 5) {
		cerr << endl << "Usage: " << argv[0] << " -f <fasta-file> -o <out-directory> -s <sam-files-glob> -c <source-sam-file>" << endl << endl;
		return -1;
	}
	string	assembly_file, out_directory, sam_glob, this_sample_str;
	for(int i=1; i+1<argc; i+=2) {				// ref :449-462
		if(!strcmp(argv[i], "-f"))		assembly_file = argv[i+1];
		else if(!strcmp(argv[i], "-o"))		out_directory = argv[i+1];
		else if(!strcmp(argv[i], "-s"))		sam_glob = argv[i+1];
		else if(!strcmp(argv[i], "-c"))		this_sample_str = argv[i+1];
		else { cerr << endl << "Unknown option " << argv[i] << endl << endl; return -1; }
	}
	if(assembly_file.empty() || out_directory.empty() || sam_glob.empty() || this_sample_str.empty()) {
		cerr << endl << "Error: -f, -o, -s and -c are all required" << endl << endl;
		return -1;
	}

	abw_ctx* ctx = NULL;
	if(abw_ctx_create(abw_env_device(), &ctx) != ABW_OK) {
		cerr << "abawaca-build (hip): no MI355X / HIP device available" << endl;
		return -1;
	}

	// ---- the assembly (ref :481-490): the reference's reader; the sequences go to the GPU in BATCHES of whole scaffolds
	// (ABW_BUILD_BATCH_MB of sequence, default 256) instead of into Scaf objects: while the reader -- one host thread, as
	// in the reference -- parses batch i + 1, a worker thread has batch i segmented and counted (abw_sig_build, ref :198-228,
	// :103-174) and counts the N of every segment for abawaca.names; then the batch's sequence is dropped.  Memory stays at
	// two batches however large the assembly is, and the GPU's work hides behind the parsing.
	cerr << time_since_epoch() << "Reading assembly file (" << assembly_file << ")" << endl;
	const double t_start = now_s();
	vector<ScafRec>		recs;
	map<string, size_t>	by_name;			// the reference's std::map order = byte-wise name order
	uint64_t		n_seg = 0;
	vector<uint32_t>	seg_scaf;
	vector<uint64_t>	seg_start, seg_end;
	vector<uint32_t>	seg_nn;				// N of every segment (abawaca.names prints "<non-N>/<size> non-Ns bps", ref :218)
	vector<int32_t>		sig_milli;			// [n_seg][179]: int(1000.0 * x) of every printed signature column
	double			gpu_busy_s = 0;
	{
		size_t batch_bytes = (size_t)256 << 20;
		if(const char* bm = getenv("ABW_BUILD_BATCH_MB")) { if(atol(bm) > 0) batch_bytes = (size_t)atol(bm) << 20; }
		if(const char* bk = getenv("ABW_BUILD_BATCH_KB")) { if(atol(bk) > 0) batch_bytes = (size_t)atol(bk) << 10; }	// the tests
		struct Batch { vector<uint8_t> blob; vector<uint64_t> off; size_t first_scaf; };
		std::thread	worker;
		string		worker_error;
		auto run_batch = [&](Batch* b) {			// on the worker thread: the only user of ctx until it is joined
			const double t0 = now_s();
			const size_t nb = b->off.size() - 1;
			uint64_t got = 0, cap = b->blob.size() / window_size + nb + 16;
			vector<uint32_t> sc; vector<uint64_t> st, en; vector<int32_t> mi;
			for(;;) {
				sc.resize(cap); st.resize(cap); en.resize(cap); mi.resize(cap * ABW_SIG_NCOLS);
				const int rc = abw_sig_build(ctx, b->blob.data(), b->off.data(), nb, (uint32_t)window_size, cap, sc.data(), st.data(), en.data(),
				                             NULL, mi.data(), &got);
				if(rc == ABW_ERR_CAPACITY) { cap = got; continue; }	// scaffolds of N only make a segment per base
				// (ABW_ERR_LOWER_N cannot happen: DNAString's constructor already threw on a lower-case n, ref src/String.cpp:45-49)
				if(rc != ABW_OK) worker_error = string("abw_sig_build: ") + abw_last_error(ctx);
				break;
			}
			if(worker_error.empty()) {
				const size_t at = seg_scaf.size();
				seg_scaf.resize(at + got); seg_start.resize(at + got); seg_end.resize(at + got); seg_nn.resize(at + got);
				sig_milli.resize((at + got) * ABW_SIG_NCOLS);
				for(uint64_t g = 0; g < got; g++) {
					seg_scaf[at + g] = sc[g] + (uint32_t)b->first_scaf; seg_start[at + g] = st[g]; seg_end[at + g] = en[g];
					const uint8_t* p = b->blob.data() + b->off[sc[g]] + st[g] - 1;
					const size_t size = (size_t)(en[g] - st[g] + 1);
					uint32_t nn = 0;
					for(size_t q = 0; q < size; q++) nn += p[q] == 'N';
					seg_nn[at + g] = nn;
				}
				memcpy(&sig_milli[at * ABW_SIG_NCOLS], mi.data(), (size_t)got * ABW_SIG_NCOLS * sizeof(int32_t));
				n_seg += got;
			}
			gpu_busy_s += now_s() - t0;
			delete b;
		};
		auto hand_over = [&](Batch* b) {
			if(worker.joinable()) worker.join();
			if(!worker_error.empty()) { cerr << "abawaca-build (hip): " << worker_error << endl; exit(-1); }
			if(b) worker = std::thread(run_batch, b);
		};
		Batch* cur = new Batch();
		cur->off.assign(1, 0); cur->first_scaf = 0;
		Bio::SeqIORead_fasta<Bio::DNASequence>	in(assembly_file);
		Bio::DNASequence*			seq = NULL;
		while((seq = in.next_seq()) != NULL) {
			const Bio::DNAString& dna = seq->seq();
			if(!by_name.insert(pair<string, size_t>(seq->display_id(), recs.size())).second) {
				cerr << "abawaca-build (hip): two sequences named " << seq->display_id() << endl;
				return -1;
			}
			ScafRec r;
			r.name = seq->display_id(); r.len = dna.size(); r.ns = Bio::Ns(dna); r.gc = Bio::gc(dna); r.nbps = 0;
			recs.push_back(r);
			cur->blob.insert(cur->blob.end(), dna.begin(), dna.end());
			cur->off.push_back(cur->blob.size());
			delete seq;
			if(cur->blob.size() >= batch_bytes) {
				hand_over(cur);
				cur = new Batch();
				cur->off.assign(1, 0); cur->first_scaf = recs.size();
			}
		}
		if(cur->off.size() > 1) hand_over(cur); else delete cur;
		hand_over(NULL);
	}
	const double t_assembly = now_s();
	const size_t n_scaf = recs.size();
	vector<uint8_t>		name_blob;				// the scaffold names as abw_sam_parse takes them, in input order
	bool				names_resident = false;	// ... and whether abw_sam_names_load has put their table on the device
	vector<uint64_t>	name_off(1, 0);
	for(size_t sc = 0; sc < n_scaf; sc++) {
		name_blob.insert(name_blob.end(), recs[sc].name.begin(), recs[sc].name.end());
		name_off.push_back(name_blob.size());
	}
	name_blob.push_back(0);

	// ---- coverage (ref :507-559): the reference's SAM reader and filter, sums on the GPU in SAM order
	vector<string>		sample_names;
	vector<vector<double> >	cov;				// [sample][n_seg]
	size_t			this_sample = (size_t)-1;
	{
		vector<string>	sam_files = glob(sam_glob);
		for(auto vit=sam_files.begin(); vit!=sam_files.end(); vit++) {
			cerr << time_since_epoch() << "Reading SAM file " << *vit << endl;
			const size_t dimension = sample_names.size();
			sample_names.push_back(*vit);
			if(this_sample_str == *vit)
				this_sample = dimension;
			cov.push_back(vector<double>(n_seg, 0.0));
			vector<uint32_t>	r_scaf;
			vector<uint64_t>	r_pos;
			vector<uint32_t>	r_len;
			auto flush = [&]() {
				if(r_scaf.empty()) return;
				if(abw_cov_accumulate(ctx, seg_scaf.data(), seg_start.data(), seg_end.data(), n_seg, n_scaf, r_scaf.data(), r_pos.data(),
				                      r_len.data(), r_scaf.size(), cov.back().data()) != ABW_OK)
					die(ctx, "abw_cov_accumulate");
				r_scaf.clear(); r_pos.clear(); r_len.clear();
			};
			// The records through abw_sam_parse (the reference's reader, record parser, filter and name lookup as kernels),
			// the file in pieces cut at line ends.  A line that parser refuses -- one the reference's reader does not survive
			// either, or one of the few things it is stricter about -- sends the WHOLE file through the reference's reader
			// below instead (ABW_SAM_READER=reference in the environment: always).
			bool parsed = false;
			const char* which = getenv("ABW_SAM_READER");
			if(!(which && !strcmp(which, "reference"))) {
				FILE* fp = fopen(vit->c_str(), "rb");
				if(fp == NULL) { cerr << "abawaca-build (hip): failed to open SAM file " << *vit << endl; return -1; }
				size_t piece = (size_t)64 << 20;				// bytes read per step (ABW_SAM_PIECE_KB: another size, for the tests)
				if(const char* pk = getenv("ABW_SAM_PIECE_KB")) { if(atol(pk) > 0) piece = (size_t)atol(pk) << 10; }
				vector<uint8_t>	buf;
				vector<size_t>	nbps_add(dimension == this_sample ? n_scaf : 0, 0);
				size_t have = 0;
				bool ok = true, eof = false;
				while(ok && !(eof && have == 0)) {
					buf.resize(have + piece);
					const size_t got = eof ? 0 : fread(buf.data() + have, 1, piece, fp);
					if(got < piece) eof = true;
					have += got;
					size_t cut = have;				// the whole rest at the end of the file, else up to the last line end
					if(!eof) { while(cut > 0 && buf[cut - 1] != '\n') cut--; if(cut == 0) continue; }
					size_t cap_r = 1;
					for(size_t i = 0; i < cut; i++) cap_r += buf[i] == '\n';
					r_scaf.resize(cap_r); r_pos.resize(cap_r); r_len.resize(cap_r);
					abw_sam_counts cnt;
					if(!names_resident) {				// once for all pieces of all SAM files
						if(abw_sam_names_load(ctx, name_blob.data(), name_off.data(), n_scaf) != ABW_OK) die(ctx, "abw_sam_names_load");
						names_resident = true;
					}
					const int rc = abw_sam_parse(ctx, buf.data(), cut, NULL, NULL, n_scaf, (uint32_t)max_snps, cap_r,
					                             r_scaf.data(), r_pos.data(), r_len.data(), NULL, NULL, NULL, NULL, NULL, &cnt);
					if(rc == ABW_ERR_FORMAT || rc == ABW_ERR_LOWER_N) {
						cerr << "abawaca-build (hip): " << abw_last_error(ctx) << endl;
						ok = false;
						break;
					}
					if(rc != ABW_OK) die(ctx, "abw_sam_parse");
					// names the assembly does not have come back as n_scaf: dropped here as the reference drops them (ref :550-552)
					size_t k = 0;
					for(size_t i = 0; i < cnt.kept; i++) {
						if(r_scaf[i] >= n_scaf) continue;
						r_scaf[k] = r_scaf[i]; r_pos[k] = r_pos[i]; r_len[k] = r_len[i];
						if(dimension == this_sample) nbps_add[r_scaf[k]] += r_len[k];		// ref :242-243
						k++;
					}
					r_scaf.resize(k); r_pos.resize(k); r_len.resize(k);
					flush();
					memmove(buf.data(), buf.data() + cut, have - cut);
					have -= cut;
				}
				fclose(fp);
				if(ok) {
					for(size_t sc = 0; sc < nbps_add.size(); sc++) recs[sc].nbps += nbps_add[sc];
					parsed = true;
				} else {
					std::fill(cov.back().begin(), cov.back().end(), 0.0);
					r_scaf.clear(); r_pos.clear(); r_len.clear();
				}
			}
			if(parsed)
				continue;
			Bio::SAMReader	reader(*vit);
			while(reader.good()) {
				Bio::ReadMappingPtr mapping(reader.next_mapping());
				if(mapping == NULL)
					break;
				if(mapping->unmapped() || (mapping->num_snps() > max_snps) || mapping->multiple_hits())	// ref :546
					continue;
				auto mit = by_name.find(mapping->ref_name());
				if(mit == by_name.end())
					continue;							// ref :550-552
				r_scaf.push_back((uint32_t)mit->second);
				r_pos.push_back((uint64_t)mapping->ref_pos());
				r_len.push_back((uint32_t)mapping->read_length());
				if(dimension == this_sample)
					recs[mit->second].nbps += mapping->read_length();		// ref :242-243
				if(r_scaf.size() >= sam_chunk)
					flush();
			}
			flush();
		}
	}
	const size_t n_samples = sample_names.size();
	const double t_sam = now_s();

	// ---- rows in the reference's order: scaffolds by name, their segments by position (ref :596-598)
	vector<uint64_t>	first_seg(n_scaf + 1, 0);			// segments arrive grouped by scaffold, in input order
	for(uint64_t g = 0; g < n_seg; g++) first_seg[seg_scaf[g] + 1]++;
	for(size_t s = 0; s < n_scaf; s++) first_seg[s + 1] += first_seg[s];
	const int32_t		n_cols = (int32_t)(ABW_SIG_NCOLS + n_samples);
	vector<int32_t>		rows((size_t)n_seg * n_cols);
	vector<uint64_t>	row_seg;
	row_seg.reserve(n_seg);
	for(auto it=by_name.begin(); it!=by_name.end(); it++)
		for(uint64_t g = first_seg[it->second]; g < first_seg[it->second + 1]; g++)
			row_seg.push_back(g);
	for(uint64_t r = 0; r < n_seg; r++) {
		const uint64_t g = row_seg[r];
		memcpy(&rows[r * n_cols], &sig_milli[g * ABW_SIG_NCOLS], ABW_SIG_NCOLS * sizeof(int32_t));
		for(size_t s = 0; s < n_samples; s++) {
			const double v = 1000.0 * cov[s][g];				// ref :603: int(1000.0 * x)
			if(!(v > -2147483648.0 && v < 2147483648.0)) { cerr << "abawaca-build (hip): coverage value out of range" << endl; return -1; }
			rows[r * n_cols + ABW_SIG_NCOLS + s] = (int32_t)v;
		}
	}

	char names_file[1024], lrn_file[1024], info_file[1024], links_file[1024];
	sprintf(names_file, "%s/abawaca.names", out_directory.c_str());
	sprintf(lrn_file, "%s/abawaca.lrn", out_directory.c_str());
	sprintf(info_file, "%s/abawaca.info", out_directory.c_str());
	sprintf(links_file, "%s/abawaca.links", out_directory.c_str());
	cerr << time_since_epoch() << "Creating .lrn (" << lrn_file << ") and .names (" << names_file << ") files" << endl;
	if(!directory_exists(out_directory.c_str())) {
		char cmd[1024];
		sprintf(cmd, "mkdir %s", out_directory.c_str());
		system(cmd);
	}
	FILE* fout = fopen(names_file, "w");
	FILE* flrn = fopen(lrn_file, "w");
	FILE* finfo = fopen(info_file, "w");
	if(fout == NULL || flrn == NULL || finfo == NULL) {
		cerr << "Could not write to " << out_directory << endl << endl;
		return -1;
	}

	// headers (ref :578-593): the dimension count includes the unprinted "A"
	vector<char>	dim_buf(1024);
	if(abw_sig_dim_names(dim_buf.data(), dim_buf.size()) != ABW_OK) die(ctx, "abw_sig_dim_names");
	vector<string>	dim_names;
	for(const char* p = dim_buf.data(); dim_names.size() < ABW_SIG_NDIMS; p += strlen(p) + 1) dim_names.push_back(p);
	fprintf(fout, "%c %lu\n", '%', (unsigned long)n_seg);
	fprintf(flrn, "%c %lu\n", '%', (unsigned long)n_seg);
	fprintf(flrn, "%c %lu\n", '%', (unsigned long)(ABW_SIG_NDIMS + n_samples));
	fprintf(flrn, "%c 9", '%');
	for(int32_t c = 0; c < n_cols; c++) fprintf(flrn, "\t1");
	fprintf(flrn, "\n");
	fprintf(flrn, "%c Key", '%');
	for(size_t d = 1; d < dim_names.size(); d++) fprintf(flrn, "\t%s", dim_names[d].c_str());
	for(size_t s = 0; s < n_samples; s++) fprintf(flrn, "\t%s", sample_names[s].c_str());
	fprintf(flrn, "\n");

	// the rows of abawaca.lrn (ref :600-605) printed on the GPU
	if(n_seg > 0) {
		uint64_t text_len = 0;
		int rc = abw_lrn_emit(ctx, rows.data(), (int64_t)n_seg, n_cols, 1, NULL, 0, &text_len);
		if(rc != ABW_ERR_CAPACITY && rc != ABW_OK) die(ctx, "abw_lrn_emit");
		vector<uint8_t> text(text_len);
		if(abw_lrn_emit(ctx, rows.data(), (int64_t)n_seg, n_cols, 1, text.data(), text.size(), &text_len) != ABW_OK) die(ctx, "abw_lrn_emit");
		fwrite(text.data(), 1, text_len, flrn);
	}

	// abawaca.info and abawaca.names (ref :596-599; the segment's name and description are those of ref :214-219)
	uint64_t i = 0;
	for(auto it=by_name.begin(); it!=by_name.end(); it++) {
		const ScafRec& r = recs[it->second];
		const double cvg = double(r.nbps) / r.len;					// Scaf::cvg, ref :57
		fprintf(finfo, "%s\t%lu\t%.3lf\t%.3lf\t%lu\n", r.name.c_str(), (unsigned long)r.len, int(1000.0*cvg)/1000.0, int(1000.0*r.gc)/1000.0, (unsigned long)r.ns);
		size_t k = 0;
		for(uint64_t g = first_seg[it->second]; g < first_seg[it->second + 1]; g++) {
			const size_t size = (size_t)(seg_end[g] - seg_start[g] + 1);
			const size_t nn = seg_nn[g];
			fprintf(fout, "%lu\t%s_%lu\t%s:(%lu, %lu), %lu/%lu non-Ns bps\n", (unsigned long)++i, r.name.c_str(), (unsigned long)++k, r.name.c_str(),
			        (unsigned long)seg_start[g], (unsigned long)seg_end[g], (unsigned long)(size - nn), (unsigned long)size);
		}
	}
	fclose(fout);
	fclose(flrn);
	fclose(finfo);

	// data.txt (ref :615-630) with the same keys; the SCG table is not predicted here (external tools, ref :401-431)
	char data[2048];
	sprintf(data, "%s/data.txt", out_directory.c_str());
	FILE* fp = fopen(data, "w");
	if(fp != NULL) {
		size_t l = assembly_file.rfind('/');
		string out_name = (l == string::npos)? assembly_file : assembly_file.substr(l+1);
		fprintf(fp, "SCG\t%s/%s.proteins.faa.bacteria.scg\n", out_directory.c_str(), out_name.c_str());
		fprintf(fp, "Links\t%s\n", links_file);
		fprintf(fp, "Info\t%s\n", info_file);
		fprintf(fp, "Names\t%s\n", names_file);
		fprintf(fp, "Lrn\t%s\n", lrn_file);
		fprintf(fp, "Assembly\t%s\n", assembly_file.c_str());
		fclose(fp);
	}
	abw_ctx_destroy(ctx);
	if(const char* tf = getenv("ABW_BUILD_TIMES")) {			// where the wall clock went (bench.py's end-to-end leg)
		FILE* ft = fopen(tf, "w");
		if(ft) {
			fprintf(ft, "{\"assembly_s\": %.4f, \"of_which_sig_build_on_the_worker_s\": %.4f, \"sam_and_coverage_s\": %.4f, \"output_s\": %.4f, "
			            "\"total_s\": %.4f, \"scaffolds\": %lu, \"segments\": %lu}\n", t_assembly - t_start, gpu_busy_s, t_sam - t_assembly,
			        now_s() - t_sam, now_s() - t_start, (unsigned long)n_scaf, (unsigned long)n_seg);
			fclose(ft);
		}
	}
	cerr << time_since_epoch() << "Finished successfully" << endl;
	return 0;
}
