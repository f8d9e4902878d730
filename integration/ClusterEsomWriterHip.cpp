// See ClusterEsomWriterHip.h.
#include "ClusterEsomWriterHip.h"
#include "abawaca_hip.h"
#include <stdlib.h>
// the GPU this process works on: ABW_DEVICE in the environment (default 0)
static inline int abw_env_device() { const char* e = getenv("ABW_DEVICE"); return e ? atoi(e) : 0; }
#include <stdexcept>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

namespace {

struct WriterStore {
	abw_ctx*		ctx = NULL;
	const ClusterData*	root = NULL;
	size_t			N = 0, D = 0;
};
WriterStore g_writer;

void load_table(const ScafDpData& scaf_db, const ClusterData& root)
{
	WriterStore& st = g_writer;
	if(!st.ctx && abw_ctx_create(abw_env_device(), &st.ctx) != ABW_OK)
		throw std::runtime_error("ClusterEsomWriterHip: no MI355X / HIP device available");
	const size_t N = scaf_db.ndps(), D = root.ndimensions();
	std::vector<double> values(D * N, 0.0);				// [D][N], column dp-1 = datapoint dp
	for(size_t d = 0; d < D; d++) {
		const Dimension& dim = root.get_dimension(d + 1);
		for(auto it = dim.begin(); it != dim.end(); it++)
			values[d * N + (it->dp - 1)] = it->value;
	}
	if(abw_lrn_table_load(st.ctx, values.data(), (int64_t)N, (int32_t)D) != ABW_OK)
		throw std::runtime_error(std::string("ClusterEsomWriterHip: abw_lrn_table_load failed: ") + abw_last_error(st.ctx));
	st.root = &root; st.N = N; st.D = D;
}

}	// namespace

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
ClusterEsomWriterHip::ClusterEsomWriterHip(string _out_dir, const ScafDpData& _scaf_db, const ClusterData& _root) :
	ClusterWriter(_out_dir, _scaf_db), root(_root)
{
	if(g_writer.ctx == NULL || g_writer.root != &_root)
		load_table(_scaf_db, _root);
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterEsomWriterHip::shutdown()
{
	if(g_writer.ctx) abw_ctx_destroy(g_writer.ctx);
	g_writer = WriterStore();
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
void ClusterEsomWriterHip::write_clustering_info_file(const ClusterData& clustering_db, const char* cluster_name) const
{
	const WriterStore& st = g_writer;
	string	lrn_file = out_dir + "/" + cluster_name + ".lrn";
	FILE*	fp = fopen(lrn_file.c_str(), "w");
	if(fp == NULL) {
		cerr << "Fatal error, " << __FILE__ << " (" << __LINE__ << "): could write to file " << lrn_file << endl << endl;
		exit(-1);
	}
	// the four header lines (same bytes as reference src/ClusterWriter.cpp:81-90; dimension names from the root table)
	std::string head = "% " + std::to_string(clustering_db.ndps()) + "\n% " + std::to_string(st.D + 1) + "\n% 9";
	for(size_t i = 0; i < st.D; i++) head += "\t1";
	head += "\n% Key";
	for(size_t i = 1; i <= st.D; i++) head += "\t" + root.get_dimension(i).get_dimension_name();
	head += "\n";
	fwrite(head.data(), 1, head.size(), fp);

	// the rows (:92-98) in the order the reference walks the cluster's datapoints
	const unordered_set<size_t>& dps = clustering_db.datapoints();
	std::vector<int64_t> list(dps.begin(), dps.end());
	// one call prints and sizes: the buffer is kept between clusters and starts at a guess of 12 bytes per value; a
	// cluster that takes more (ABW_ERR_CAPACITY carries the size) is printed again into a buffer of that size
	static std::vector<uint8_t> text;
	const uint64_t guess = (uint64_t)list.size() * ((uint64_t)st.D * 12 + 24);
	if(text.size() < guess) text.resize(guess);
	uint64_t len = 0;
	int rc = list.empty() ? ABW_OK : abw_lrn_emit_rows(st.ctx, NULL, (int64_t)st.N, (int32_t)st.D, list.data(), (int64_t)list.size(), text.data(), text.size(), &len);
	if(rc == ABW_ERR_CAPACITY) {
		text.resize(len);
		rc = abw_lrn_emit_rows(st.ctx, NULL, (int64_t)st.N, (int32_t)st.D, list.data(), (int64_t)list.size(), text.data(), text.size(), &len);
	}
	if(rc != ABW_OK)
		throw std::runtime_error(std::string("ClusterEsomWriterHip: abw_lrn_emit_rows failed: ") + abw_last_error(st.ctx));
	fwrite(text.data(), 1, len, fp);
	fclose(fp);
}
