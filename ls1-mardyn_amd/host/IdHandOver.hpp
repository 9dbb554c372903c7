// IdHandOver.hpp — how rank 0 hands a small opaque payload (the ncclUniqueId, 128 bytes) to the other ranks of ONE job on one
// node before any communicator exists.  Pure POSIX, no HIP / RCCL: tests/hostcpp/id_handover_test.cpp runs it on the CPU.
//
// What it stands in for in the reference: MPI_Init / MPI_COMM_WORLD bring the ranks of a job together there
// (parallel/DomainDecompMPIBase.cpp); a non-MPI build launched one-process-per-GPU has only the launcher's environment.
//
//   id_source = "tcp:<host>:<port>"  rank 0 listens there and sends the payload to every rank that connects and names itself
//               (the default of DomainDecompHip: the launcher's MASTER_ADDR and MASTER_PORT + 17).  Nothing survives a job: a stale
//               hand-over from a crashed or concurrent run cannot be picked up; a port that is taken is an error, not a mix-up.
//   id_source = <path>               rank 0 publishes the payload in that file (LS1HIP_RCCL_ID_FILE: hosts without a usable port).
//               The file carries a magic word and rank 0's clock; readers reject a file older than 120 s as a left-over of an
//               earlier run, rank 0 replaces an existing file (and removes its own once the communicator is up: the caller does).
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace ls1hip {

constexpr int ID_FILE_MAX_AGE_S = 120;

inline void hand_over_file(int rank, const std::string& id_file, void* payload, size_t bytes, int wait_ms = 60000) {
	struct Head {
		char magic[8];
		int64_t unix_seconds;
		uint64_t bytes;
	};
	if (rank == 0) {
		Head h;
		std::memcpy(h.magic, "LS1RCCL1", 8);
		h.unix_seconds = (int64_t)time(nullptr);
		h.bytes = bytes;
		(void)remove(id_file.c_str());  // left over from an earlier run
		const std::string tmp = id_file + ".tmp";
		FILE* f = fopen(tmp.c_str(), "wb");
		const bool ok = f && fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(payload, bytes, 1, f) == 1;
		if (f && fclose(f) != 0) throw std::runtime_error("cannot write " + tmp);
		if (!ok) throw std::runtime_error("cannot write " + tmp);
		if (rename(tmp.c_str(), id_file.c_str())) throw std::runtime_error("cannot publish " + id_file);
		return;
	}
	std::vector<char> buf(bytes);
	for (int waited = 0;; waited += 10) {
		FILE* f = fopen(id_file.c_str(), "rb");
		if (f) {
			Head h;
			const bool ok = fread(&h, sizeof(h), 1, f) == 1 && h.bytes == bytes && fread(buf.data(), bytes, 1, f) == 1;
			fclose(f);
			const int64_t age = (int64_t)time(nullptr) - h.unix_seconds;
			if (ok && !std::memcmp(h.magic, "LS1RCCL1", 8) && age >= -5 && age <= ID_FILE_MAX_AGE_S) {
				std::memcpy(payload, buf.data(), bytes);
				return;
			}
		}
		if (waited > wait_ms)
			throw std::runtime_error("no fresh hand-over record in " + id_file + " (a file older than " + std::to_string(ID_FILE_MAX_AGE_S) +
									 " s is taken for a left-over)");
		std::this_thread::sleep_for(std::chrono::milliseconds(10));
	}
}

inline void hand_over_tcp(int world, int rank, const std::string& host_port, void* payload, size_t bytes, int wait_ms = 120000) {
	const size_t c0 = host_port.rfind(':');
	if (c0 == std::string::npos) throw std::runtime_error("bad tcp hand-over address '" + host_port + "' (host:port)");
	std::string host = host_port.substr(0, c0);
	const int port = atoi(host_port.c_str() + c0 + 1);
	if (port <= 0 || port > 65535) throw std::runtime_error("bad tcp hand-over port in '" + host_port + "'");
	sockaddr_in a;
	std::memset(&a, 0, sizeof(a));
	a.sin_family = AF_INET;
	a.sin_port = htons((uint16_t)port);
	if (rank == 0) {
		const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
		if (ls < 0) throw std::runtime_error("socket() failed");
		int one = 1;
		(void)::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
		a.sin_addr.s_addr = htonl(INADDR_ANY);
		if (::bind(ls, reinterpret_cast<sockaddr*>(&a), sizeof(a)) != 0 || ::listen(ls, world) != 0) {
			(void)::close(ls);
			throw std::runtime_error("cannot listen on port " + std::to_string(port) + " for the hand-over (taken by another job?  set "
									 "LS1HIP_RCCL_PORT or LS1HIP_RCCL_ID_FILE)");
		}
		std::vector<char> seen((size_t)world, 0);
		for (int served = 0; served < world - 1;) {
			pollfd pf = {ls, POLLIN, 0};
			if (::poll(&pf, 1, wait_ms) <= 0) {
				(void)::close(ls);
				throw std::runtime_error("only " + std::to_string(served) + " of " + std::to_string(world - 1) + " ranks asked for the hand-over in time");
			}
			const int c = ::accept(ls, nullptr, nullptr);
			if (c < 0) continue;
			int32_t who = -1;
			if (::recv(c, &who, sizeof(who), MSG_WAITALL) == (ssize_t)sizeof(who) && who > 0 && who < world &&
				::send(c, payload, bytes, MSG_NOSIGNAL) == (ssize_t)bytes && !seen[(size_t)who]) {
				seen[(size_t)who] = 1;
				++served;
			}
			(void)::close(c);
		}
		(void)::close(ls);
		return;
	}
	if (host.empty() || host == "localhost") host = "127.0.0.1";
	if (inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) throw std::runtime_error("tcp hand-over needs a numeric IPv4 address, got '" + host + "'");
	for (int waited = 0;; waited += 100) {
		const int c = ::socket(AF_INET, SOCK_STREAM, 0);
		if (c < 0) throw std::runtime_error("socket() failed");
		if (::connect(c, reinterpret_cast<sockaddr*>(&a), sizeof(a)) == 0) {
			const int32_t who = rank;
			const bool ok = ::send(c, &who, sizeof(who), MSG_NOSIGNAL) == (ssize_t)sizeof(who) && ::recv(c, payload, bytes, MSG_WAITALL) == (ssize_t)bytes;
			(void)::close(c);
			if (ok) return;
		} else {
			(void)::close(c);
		}
		if (waited > wait_ms) throw std::runtime_error("rank 0 does not answer on " + host_port + " (hand-over)");
		std::this_thread::sleep_for(std::chrono::milliseconds(100));
	}
}

// id_source: "tcp:<host>:<port>" or a file path (see the head of this file)
inline void hand_over_bytes(int world, int rank, const std::string& id_source, void* payload, size_t bytes) {
	if (world <= 1) return;
	if (id_source.rfind("tcp:", 0) == 0) hand_over_tcp(world, rank, id_source.substr(4), payload, bytes);
	else hand_over_file(rank, id_source, payload, bytes);
}

// Where the hand-over happens, from the launcher's environment: LS1HIP_RCCL_ID_FILE = an explicit file; otherwise a TCP rendezvous on
// MASTER_ADDR : (LS1HIP_RCCL_PORT, or MASTER_PORT + 17) — per job by construction (the launcher's port is the job's), nothing on
// disk.  There is no fixed path in /tmp: a stale or foreign file there could be taken for this job's.
inline std::string rccl_id_source_from_env() {
	if (const char* f = getenv("LS1HIP_RCCL_ID_FILE")) return f;
	const char* addr = getenv("MASTER_ADDR");
	int port = -1;
	if (const char* p = getenv("LS1HIP_RCCL_PORT")) port = atoi(p);
	else if (const char* mp = getenv("MASTER_PORT")) port = atoi(mp) + 17;
	if (port <= 0 || port > 65535)
		throw std::runtime_error("multi-rank RCCL transport: no rendezvous for the ncclUniqueId — launch with MASTER_ADDR / MASTER_PORT "
								 "(torchrun-style), or set LS1HIP_RCCL_PORT, or LS1HIP_RCCL_ID_FILE");
	return std::string("tcp:") + (addr ? addr : "127.0.0.1") + ":" + std::to_string(port);
}

}  // namespace ls1hip
