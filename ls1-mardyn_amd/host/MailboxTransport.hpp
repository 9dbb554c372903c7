// MailboxTransport.hpp — host-staged transport for the decomposed loop: messages are files in a directory all ranks share
// (tmpfs: /dev/shm), written under a temporary name and published by rename(2), polled by the receiver.
//
// What it is for: several ranks of the decomposed loop on ONE GPU (RCCL refuses two ranks on one device) — the one-GPU box of
// the -m gpu tests, where the multi-rank seam of the reference driver (DomainDecompHip) is checked with 2 and 4 MarDyn
// processes — and hosts without a fabric.  Device buffers are staged through the host (hipMemcpy).  Throughput is irrelevant
// here; the protocol above it (merged per-peer messages, count exchange with a status word, collective errors) is exactly the
// one the RCCL transport carries (DomainDecompRccl.hpp), which is the production transport.
//
// Interface = the duck-typed transport of DomainDecompRccl.hpp, plus all_gather_bytes for the reference's typed collectives.
#pragma once
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace ls1hip {

class MailboxTransport {
public:
	MailboxTransport(int world, int rank, const std::string& dir, double timeout_s = 300.) : _world(world), _rank(rank), _dir(dir), _timeout(timeout_s) {
		if (_dir.empty()) throw std::runtime_error("MailboxTransport: no directory (LS1HIP_COMM_DIR)");
		mkdir(_dir.c_str(), 0700);  // (exists already for all ranks but the first)
		// Messages are keyed by (job key, tag, src, dst, seq) with seq starting at 0 in every process: files a crashed or killed
		// earlier run left in the directory must not be taken for this run's.  The job key comes from the launcher's environment,
		// which every rank of a job shares without talking (LS1HIP_JOB_KEY, else the launcher's run id / rendezvous port); give
		// every run a fresh directory where the launcher sets none of them (the tests do: mkdtemp).
		for (const char* k : {"LS1HIP_JOB_KEY", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "MASTER_PORT"})
			if (const char* e = getenv(k)) {
				for (const char* c = e; *c; ++c) _key += (isalnum((unsigned char)*c) ? *c : '-');
				break;
			}
	}
	int world() const { return _world; }
	int rank() const { return _rank; }

	// [world][n] table of every rank's n bytes
	void all_gather_bytes(const void* mine, size_t n, std::vector<char>& all) {
		all.assign((size_t)_world * n, 0);
		std::memcpy(all.data() + (size_t)_rank * n, mine, n);
		const uint64_t seq = _coll_seq++;
		for (int r = 0; r < _world; ++r)
			if (r != _rank) put("ag", r, seq, mine, n);
		for (int r = 0; r < _world; ++r)
			if (r != _rank) get("ag", r, seq, all.data() + (size_t)r * n, n);
	}
	std::vector<int64_t> all_gather(const int64_t* mine, int n) {
		std::vector<char> all;
		all_gather_bytes(mine, (size_t)n * sizeof(int64_t), all);
		std::vector<int64_t> out((size_t)_world * n);
		std::memcpy(out.data(), all.data(), all.size());
		return out;
	}
	template <class OP>
	void all_reduce(double* v, int n, OP op) {  // OP: ReduceOp of DomainDecompRccl.hpp (Sum = 0, Max = 1); summed in rank order on every rank
		if (_world == 1) return;
		std::vector<char> all;
		all_gather_bytes(v, (size_t)n * sizeof(double), all);
		const double* t = reinterpret_cast<const double*>(all.data());
		for (int k = 0; k < n; ++k) {
			double acc = t[k];
			for (int r = 1; r < _world; ++r) {
				const double x = t[(size_t)r * n + k];
				acc = static_cast<int>(op) == 0 ? acc + x : (x > acc ? x : acc);
			}
			v[k] = acc;
		}
	}
	void barrier() {
		char c = 0;
		std::vector<char> all;
		all_gather_bytes(&c, 1, all);
	}

	// grouped point-to-point of device buffers: the sends are posted at once, the receives complete in group_end
	void group_start() { _pending.clear(); }
	void send(const double* dev, size_t count, int peer) {
		std::vector<double> h(count);
		if (hipMemcpy(h.data(), dev, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("MailboxTransport: D2H copy failed");
		put("p2p", peer, _send_seq[peer]++, h.data(), count * sizeof(double));
	}
	void recv(double* dev, size_t count, int peer) { _pending.push_back({dev, count, peer}); }
	void group_end() {
		for (const Pending& p : _pending) {
			std::vector<double> h(p.count);
			get("p2p", p.peer, _recv_seq[p.peer]++, h.data(), p.count * sizeof(double));
			if (hipMemcpy(p.dev, h.data(), p.count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("MailboxTransport: H2D copy failed");
		}
		_pending.clear();
	}

private:
	struct Pending {
		double* dev;
		size_t count;
		int peer;
	};
	std::string name(const char* tag, int src, int dst, uint64_t seq) const {
		return _dir + "/" + (_key.empty() ? std::string() : _key + "_") + tag + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq);
	}
	void put(const char* tag, int dst, uint64_t seq, const void* data, size_t n) {
		const std::string fin = name(tag, _rank, dst, seq), tmp = fin + ".tmp";
		FILE* f = fopen(tmp.c_str(), "wb");
		const bool ok = f && (!n || fwrite(data, 1, n, f) == n);
		if (f && fclose(f) != 0) throw std::runtime_error("MailboxTransport: cannot write " + tmp + " (directory full?)");
		if (!ok) throw std::runtime_error("MailboxTransport: cannot write " + tmp);
		if (rename(tmp.c_str(), fin.c_str())) throw std::runtime_error("MailboxTransport: cannot publish " + fin);
	}
	void get(const char* tag, int src, uint64_t seq, void* data, size_t n) {
		const std::string fin = name(tag, src, _rank, seq);
		const auto t0 = std::chrono::steady_clock::now();
		for (long spin = 0;; ++spin) {
			FILE* f = fopen(fin.c_str(), "rb");
			if (f) {
				const size_t got = n ? fread(data, 1, n, f) : 0;
				fclose(f);
				if (got != n) throw std::runtime_error("MailboxTransport: message " + fin + " has the wrong size");
				unlink(fin.c_str());
				return;
			}
			if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(200));
			if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > _timeout)
				throw std::runtime_error("MailboxTransport: no message " + fin + " (peer dead?)");
		}
	}

	int _world, _rank;
	std::string _dir, _key;
	double _timeout;
	uint64_t _coll_seq = 0;
	std::map<int, uint64_t> _send_seq, _recv_seq;
	std::vector<Pending> _pending;
};

}  // namespace ls1hip
