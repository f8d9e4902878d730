// ClusterDataView — a node's ClusterData WITHOUT its per-dimension tables.
//
// The reference builds, for every node of the recursion, a full column store of the node's datapoints
// (reference src/abawaca.cpp:101 -> src/ClusterData.cpp:10-16 -> src/Dimension.cpp:22-27: per dimension and datapoint a
// map lookup, a set insert, a map insert and a push_back) -- about half of its wall time once separate() itself is on
// the GPU.  ClusterSeparatorHip / ClusterSeparatorKMeans keep the whole table resident on the device and read from a
// node's ClusterData only datapoints() (reference src/ClusterData.h:29); this subclass gives them exactly that.  It is
// built through the reference's own subset constructor with an empty subset (D empty Dimension objects) and then
// takes the node's datapoints.  Use it for every node except the first one the separator sees (that one is the source
// of the resident table), and only where nothing else reads the node's values -- the reference's ClusterEsomWriter
// does (reference src/ClusterWriter.cpp:69-100), so with the writer enabled the reference's line stays as it is.
#ifndef CLUSTERDATAVIEW_H
#define CLUSTERDATAVIEW_H

#include "ClusterData.h"

class ClusterDataView : public ClusterData {
public:
	ClusterDataView(const ClusterData& parent, const unordered_set<size_t>& subset_dps) :
		ClusterData(parent, unordered_set<size_t>())
	{
		this->dps = subset_dps;
	}
};

#endif	// CLUSTERDATAVIEW_H
