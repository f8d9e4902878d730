// See ClusterSeparatorKMeans.h.  Host-side adapter between the reference's data model and
// abw_km_run (include/abawaca_hip.h).
#include "ClusterSeparatorKMeans.h"
#include "abawaca_hip.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct KmStore {
	abw_ctx*		ctx = NULL;
	const ScafDpData*	scaf_db = NULL;
	size_t			N = 0, D = 0;
	std::vector<float>	x;		// [D][N] z-scored over the root, dp id - 1 is the column
	int			restarts = 16, iterations = 20;
	unsigned		base_seed = 20240;
};
KmStore g_km;

void km_fail(const std::string& what)
{
	std::string msg = what;
	if(g_km.ctx) msg += std::string(": ") + abw_last_error(g_km.ctx);
	throw std::runtime_error(msg);
}

// features of every datapoint, once: the root ClusterData through its public interface
// (ClusterData::get_dimension -> Dimension::begin/end, reference src/Dimension.h:33-37)
void km_load(const ScafDpData& scaf_db, const ClusterData& root)
{
	KmStore& st = g_km;
	if(!st.ctx && abw_ctx_create(0, &st.ctx) != ABW_OK)
		throw std::runtime_error("ClusterSeparatorKMeans: no MI355X / HIP device available");
	const size_t N = scaf_db.ndps(), D = root.ndimensions();
	if(root.ndps() != N)
		throw std::invalid_argument("ClusterSeparatorKMeans: the first node must hold every datapoint (root of the recursion)");
	if(D > 223)
		throw std::invalid_argument("ClusterSeparatorKMeans: more than 223 dimensions");
	st.scaf_db = &scaf_db; st.N = N; st.D = D;
	st.x.assign(D * N, 0.0f);
	for(size_t d = 0; d < D; d++) {
		const Dimension& dim = root.get_dimension(d + 1);
		double sum = 0, sq = 0;
		for(auto it = dim.begin(); it != dim.end(); it++) { sum += it->value; sq += it->value * it->value; }
		const double mean = sum / (double)N;
		const double var = std::max(0.0, sq / (double)N - mean * mean);
		const double inv = 1.0 / std::max(1e-6, std::sqrt(var));
		for(auto it = dim.begin(); it != dim.end(); it++)
			st.x[d * N + (it->dp - 1)] = (float)((it->value - mean) * inv);
	}
}

}	// namespace

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
ClusterSeparatorKMeans::ClusterSeparatorKMeans(const ScafDpData& scaf_db, const SCGdb& scg_db, const ClusterData& clustering_db, Semaphore& threads_semaphore) :
	ClusterSeparator(scaf_db, scg_db, clustering_db, threads_semaphore, new Result())
{}

void ClusterSeparatorKMeans::set_parameters(int restarts, int iterations, unsigned base_seed)
{
	g_km.restarts = restarts; g_km.iterations = iterations; g_km.base_seed = base_seed;
}

void ClusterSeparatorKMeans::shutdown()
{
	if(g_km.ctx) abw_ctx_destroy(g_km.ctx);
	g_km = KmStore();
}

ostream& operator << (ostream& os, ClusterSeparatorKMeans& cs)
{
	os << " Best separation:\t";
	cs.print_separation_params(os);
	return os;
}

//////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////////
bool ClusterSeparatorKMeans::separate()
{
	// reference src/ClusterSeparator.cpp:59-67
	if(this->cluster1 != NULL) delete(this->cluster1);
	if(this->cluster2 != NULL) delete(this->cluster2);
	this->cluster1 = this->cluster2 = NULL;
	this->best_separation->reset();
	this->raw_dps_cluster1.clear();
	this->raw_dps_cluster2.clear();

	if(g_km.ctx == NULL || g_km.scaf_db != &scaf_db)
		km_load(scaf_db, clustering_db);
	KmStore& st = g_km;

	// the node's datapoints in ascending id order = the columns handed to the library
	std::vector<size_t> dps(clustering_db.datapoints().begin(), clustering_db.datapoints().end());
	std::sort(dps.begin(), dps.end());
	const size_t n = dps.size();
	if(n < 2 * this->cluster_ndps_threshold || n < 32)
		return false;								// two children of cluster_ndps_threshold datapoints cannot exist
	std::vector<float> x(st.D * n);
	for(size_t d = 0; d < st.D; d++)
		for(size_t i = 0; i < n; i++)
			x[d * n + i] = st.x[d * st.N + (dps[i] - 1)];

	const int R = st.restarts;
	std::vector<uint8_t> assign((size_t)R * n);
	std::vector<double> inertia(R);
	if(abw_km_run(st.ctx, x.data(), (int64_t)n, (int32_t)st.D, R, st.base_seed, st.iterations, 0, assign.data(), NULL, inertia.data()) != ABW_OK)
		km_fail("ClusterSeparatorKMeans: abw_km_run failed");
	size_t best = 0;
	for(int r = 1; r < R; r++) if(inertia[r] < inertia[best]) best = r;		// ties: the earlier restart
	const uint8_t* side = assign.data() + best * n;

	// raw split; the smaller side is cluster1 (reference :49-53; every datapoint is "useful": ScafDpData keeps
	// no single-datapoint scaffolds, reference src/ScafDpData.cpp:92-93)
	size_t n_side1 = 0;
	for(size_t i = 0; i < n; i++) n_side1 += side[i];
	const uint8_t first = (n_side1 < n - n_side1) ? 1 : 0;				// library side that becomes cluster1
	for(size_t i = 0; i < n; i++)
		(side[i] == first ? raw_dps_cluster1 : raw_dps_cluster2).insert(dps[i]);
	if(raw_dps_cluster1.size() < this->cluster_ndps_threshold) {
		raw_dps_cluster1.clear(); raw_dps_cluster2.clear();
		return false;								// no significant separation (reference :83-84)
	}

	// scaffold assignment (reference :94-101): cluster1 takes a scaffold with at least half of its datapoints there
	std::map<size_t, size_t> in1, total;
	for(size_t i = 0; i < n; i++) {
		const size_t s = scaf_db.dp2scaf(dps[i]);
		total[s]++;
		if(side[i] == first) in1[s]++;
	}
	this->cluster1 = new Cluster(scaf_db);
	this->cluster2 = new Cluster(scaf_db);
	std::map<size_t, bool> to1;
	size_t agree = 0;
	for(auto it = total.begin(); it != total.end(); it++) {
		const size_t k = in1.count(it->first) ? in1[it->first] : 0;
		const bool c1 = 2 * k >= scaf_db.ndps(it->first);
		to1[it->first] = c1;
		agree += c1 ? k : it->second - k;
		(c1 ? this->cluster1 : this->cluster2)->add_assigned_scaf(it->first);
	}
	for(size_t i = 0; i < n; i++)							// datapoints follow their scaffold (reference :103-122)
		*(to1[scaf_db.dp2scaf(dps[i])] ? this->cluster1 : this->cluster2) += dps[i];

	const double agreement = (double)agree / (double)n;
	const bool big_enough = this->cluster1->ndps() >= this->cluster_ndps_threshold && this->cluster2->ndps() >= this->cluster_ndps_threshold;
	const bool legal = big_enough && agreement >= 0.8 && cq.is_split_better(*this->cluster1, *this->cluster2);
	static_cast<Result*>(best_separation)->set(best, agreement, inertia[best], legal);
	if(!legal) {									// reference :125-132
		delete(this->cluster1); delete(this->cluster2);
		this->cluster1 = this->cluster2 = NULL;
		return false;
	}
	return true;
}
