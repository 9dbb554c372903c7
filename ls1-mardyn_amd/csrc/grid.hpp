// grid.hpp — linked-cell grid geometry shared by host and device code.
//
// Same construction as LinkedCells::rebuild (/root/reference/src/particleContainer/LinkedCells.cpp:136-204):
// cells per dimension = floor(L / (float)(rc / cellsInCutoff)), cell length = L / cells, a halo layer of
// `hw` = cellsInCutoff cells on every side, linear index x-fastest (LinkedCells.cpp:878-880), and the cell of a
// point as in getCellIndexOfPoint (:830-876) with the border snap of CellBorderAndFlagManager.h:114-130
// (a point inside the box is never binned into a halo cell and vice versa).
#pragma once
#include "pairphys.hpp"
#include <stdint.h>

namespace ls1 {

struct Grid {
	int dims[3];   // cells per dimension incl. halo
	int box[3];    // inner+boundary cells per dimension
	int hw;        // halo width in cells (= cells in cutoff)
	int ncells;
	double bmin[3], bmax[3], clen[3], crec[3];
};

// align (optional, list mode): brick edge per dimension — the cell count is rounded DOWN to a multiple of it (cells only get
// longer, never shorter than the cutoff) when that costs at most 1.5 % in cell length: partial bricks at the upper faces
// otherwise run half-empty workgroups (measured on the 10^8 box: 186 -> 184 cells per dimension, 46 full bricks instead of
// 46.5: 7.36 -> 7.68e9 updates/s; on the 10^7 box 86 -> 84 cells makes the cells 2.4 % longer and the staged regions 7 %
// larger, which costs more than the half-empty bricks: 7.42 -> 6.86e9, hence the limit)
inline bool grid_init(Grid& g, const double bmin[3], const double bmax[3], double cutoff, int cells_in_cutoff,
					  const int* align = nullptr) {
	const float rc = (float)(cutoff / cells_in_cutoff);  // float on purpose, LinkedCells.cpp:152
	g.hw = cells_in_cutoff;
	long n = 1;
	for (int d = 0; d < 3; ++d) {
		g.bmin[d] = bmin[d];
		g.bmax[d] = bmax[d];
		g.box[d] = (int)floor((bmax[d] - bmin[d]) / rc);
		if (align && align[d] > 1) {
			const int a = (g.box[d] / align[d]) * align[d];
			if (a >= align[d] && (double)a >= 0.985 * (double)g.box[d]) g.box[d] = a;
		}
		if (g.box[d] < 1) return false;  // reference: "region too small" -> exit(1), LinkedCells.cpp:161-164
		g.dims[d] = g.box[d] + 2 * g.hw;
		const double diff = bmax[d] - bmin[d];
		g.clen[d] = diff / g.box[d];
		g.crec[d] = g.box[d] / diff;
		n *= g.dims[d];
	}
	if (n > 0x7fffffffL) return false;
	g.ncells = (int)n;
	return true;
}

// cell coordinate of x in dimension d for an OWNED molecule (x in [bmin,bmax)): always an inner/boundary cell
LS1_HD int cell_coord_owned(const Grid& g, int d, double x) {
	int c = (int)floor((x - g.bmin[d]) * g.crec[d]);
	if (c < 0) c = 0;
	if (c > g.box[d] - 1) c = g.box[d] - 1;
	return c + g.hw;
}
// cell coordinate of a halo copy's shifted coordinate (may be inside or outside the box in this dimension)
LS1_HD int cell_coord_any(const Grid& g, int d, double x) {
	if (x >= g.bmin[d] && x < g.bmax[d]) return cell_coord_owned(g, d, x);
	int c = (int)floor((x - g.bmin[d]) * g.crec[d]);
	if (x < g.bmin[d]) {
		if (c > -1) c = -1;
		if (c < -g.hw) c = -g.hw;
	} else {
		if (c < g.box[d]) c = g.box[d];
		if (c > g.box[d] + g.hw - 1) c = g.box[d] + g.hw - 1;
	}
	return c + g.hw;
}
LS1_HD int cell_index(const Grid& g, int cx, int cy, int cz) { return (cz * g.dims[1] + cy) * g.dims[0] + cx; }
LS1_HD void cell_coords(const Grid& g, int c, int& cx, int& cy, int& cz) {
	cx = c % g.dims[0];
	const int t = c / g.dims[0];
	cy = t % g.dims[1];
	cz = t / g.dims[1];
}
LS1_HD bool cell_is_halo(const Grid& g, int cx, int cy, int cz) {
	return cx < g.hw || cy < g.hw || cz < g.hw || cx >= g.dims[0] - g.hw || cy >= g.dims[1] - g.hw ||
		   cz >= g.dims[2] - g.hw;
}

}  // namespace ls1
