// ClusterEsomWriterHip — the reference-side binding of libabawaca_hip.so's cluster writer.
//
// A second descendant of the reference's ClusterWriter (reference src/ClusterWriter.h:18-29) beside ClusterEsomWriter
// (:32-38): same three output files per cluster, but the data rows of <cluster>.lrn (reference
// src/ClusterWriter.cpp:92-98, one fprintf per value) are printed on the GPU from a copy of the ROOT table that stays
// on the device (abw_lrn_table_load / abw_lrn_emit_rows).  It reads from the cluster's ClusterData only ndps() and
// datapoints(), so the cluster can be a ClusterDataView.  In main the change is the type name and one more
// constructor argument at reference src/abawaca.cpp:161.  It includes the reference's headers; it contains none of the
// reference's code.
#ifndef CLUSTERESOMWRITERHIP_H
#define CLUSTERESOMWRITERHIP_H

#include "ClusterWriter.h"

class ClusterEsomWriterHip : public ClusterWriter {
public:
	// root: the table every cluster is a subset of (the ClusterData loaded from the .lrn, reference src/abawaca.cpp:86)
	ClusterEsomWriterHip(string _out_dir, const ScafDpData& _scaf_db, const ClusterData& root);
	// Releases the GPU context and the resident table (optional; call once at program end).
	static void	shutdown();
protected:
	void	write_clustering_info_file(const ClusterData& clustering_db, const char* cluster_name) const;
	const ClusterData&	root;
};

#endif	// CLUSTERESOMWRITERHIP_H
