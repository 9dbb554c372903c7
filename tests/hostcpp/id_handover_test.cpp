// CPU test driver of ls1-mardyn_amd/host/IdHandOver.hpp (the ncclUniqueId hand-over of the multi-rank C++ hosts): one process per
// rank; rank 0 owns a 128-byte payload derived from <seed>, every rank prints the payload it ends up with as hex.
//   id_handover_test <world> <rank> <id_source> <seed>
#include <cstdio>
#include <cstdlib>
#include <exception>

#include "IdHandOver.hpp"

int main(int argc, char** argv) {
	if (argc < 5) return 2;
	const int world = atoi(argv[1]), rank = atoi(argv[2]);
	unsigned char buf[128];
	unsigned s = (unsigned)atoi(argv[4]);
	for (int i = 0; i < 128; ++i) {
		s = s * 1664525u + 1013904223u;
		buf[i] = rank == 0 ? (unsigned char)(s >> 24) : 0;
	}
	try {
		ls1hip::hand_over_bytes(world, rank, argv[3], buf, sizeof(buf));
	} catch (const std::exception& e) {
		fprintf(stderr, "rank %d: %s\n", rank, e.what());
		return 1;
	}
	for (int i = 0; i < 128; ++i) printf("%02x", buf[i]);
	printf("\n");
	return 0;
}
