// ring_standin_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl that lets SEVERAL PROCESSES form a ring on ONE GPU.
//
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the development box has one GPU, so the product's
// multi-rank protocol (halo exchange order, exchange-cycle agreement, the error norm's all-reduce, the resume vote) had only
// ever met world size 1 on hardware.  This library exports the eleven entry points libcrd binds (crd_halo.cpp, RcclApi::load)
// and moves the bytes through a POSIX shared-memory segment instead of xGMI.  libcrd binds it when crd_comm_set_rccl_library names
// it (the Python package: CRD_RCCL_LIBRARY in the environment); nothing else changes: every kernel, stream, event and host decision of the ring path is the product's.
//
// Semantics kept from NCCL: the k-th send from rank a to rank b pairs with the k-th receive at b from a; the operations of a
// group complete together; an all-reduce is one call per rank in the same order; everything is ordered behind the work already
// on the stream it is given; with CRD_STANDIN_SHUFFLE the operations of a group complete in a random order, with random latencies
// (run_group_shuffled).  Not kept: asynchrony -- a call returns when its data has moved (the stream is synchronised), so a
// mis-paired protocol shows as a timeout error (CRD_STANDIN_TIMEOUT_S, default 60 s) rather than as a hang.  Sums are taken
// in rank order (RCCL's ring order differs: states are compared to round-off where a sum over ranks is involved).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kSlots = 8;            // messages in flight per (source, destination) pair: a group's sends to one peer
constexpr size_t kReduceBytes = 512;  // largest all-reduce payload

struct Channel {
	std::atomic<uint64_t> sent, consumed;
	size_t bytes[kSlots];
};

struct Shared {
	std::atomic<int> joined;
	std::atomic<uint64_t> reduce_seq[kMaxRanks];
	alignas(64) unsigned char reduce_data[2][kMaxRanks][kReduceBytes];
	Channel channel[kMaxRanks][kMaxRanks];
	// followed by the message slots: [source][destination][slot][slot_bytes]
};

size_t slot_bytes()
{
	const char *e = std::getenv("CRD_STANDIN_SLOT_MB");
	return (size_t)(e ? std::atoi(e) : 4) << 20;
}

double timeout_s()
{
	const char *e = std::getenv("CRD_STANDIN_TIMEOUT_S");
	return e ? std::atof(e) : 60.0;
}

struct Op {
	bool send;
	void *buf;
	size_t bytes;
	int peer;
	ncclComm *comm;
	hipStream_t stream;
};

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_group_ops;
thread_local char g_last_error[256] = "";

}  // namespace

struct ncclComm {
	Shared *sh = nullptr;
	unsigned char *slots = nullptr;
	size_t map_bytes = 0, slot = 0;
	int rank = 0, n = 0;
	uint64_t n_sent[kMaxRanks] = {}, n_received[kMaxRanks] = {}, n_reduced = 0;
	char name[128] = {};
	unsigned char *slot_of(int src, int dst, uint64_t seq) const { return slots + ((((size_t)src * kMaxRanks + (size_t)dst) * kSlots + (size_t)(seq % kSlots)) * slot); }
};

namespace {

ncclResult_t failed(const char *what)
{
	std::snprintf(g_last_error, sizeof g_last_error, "%s", what);
	std::fprintf(stderr, "ring_standin_rccl: %s\n", what);
	return ncclInternalError;
}

template <typename F>
bool wait_until(F &&ready)
{
	const auto t0 = std::chrono::steady_clock::now();
	const double limit = timeout_s();
	for (unsigned spin = 0; !ready(); spin++) {
		if (spin > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
		if ((spin & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
	}
	return true;
}

size_t size_of(ncclDataType_t t)
{
	switch (t) {
	case ncclDouble: case ncclInt64: case ncclUint64: return 8;
	case ncclFloat: case ncclInt32: case ncclUint32: return 4;
	case ncclInt8: case ncclUint8: return 1;
	default: return 0;
	}
}

// CRD_STANDIN_SHUFFLE=seed (round 6): the operations of a group complete in ANY order -- what NCCL promises is only that the k-th send
// from a to b pairs with the k-th receive at b from a.  The group's operations are queued per (peer, direction), in issue order; the
// progress loop below picks a queue at random, completes its head if it can go (a free slot for a send, a message there for a receive)
// and sleeps a random time of up to CRD_STANDIN_LATENCY_US microseconds in front of it.  A protocol that leans on the order in which
// this stand-in -- or an MPI -- happens to deliver (the reference's Exchange() does at two ranks, where both neighbours are one peer:
// src/FHNmodel_torus.cpp:805,811) mis-pairs its rows under it; libcrd's halo plan must not.
ncclResult_t run_group_shuffled(std::vector<Op> &ops, unsigned seed)
{
	static thread_local uint64_t state = 0;
	if (state == 0) state = 0x9e3779b97f4a7c15ull * (uint64_t)(seed + 1) + (uint64_t)getpid();
	auto rnd = [&]() {  // xorshift64*
		state ^= state >> 12;
		state ^= state << 25;
		state ^= state >> 27;
		return state * 0x2545f4914f6cdd1dull;
	};
	const char *lat = std::getenv("CRD_STANDIN_LATENCY_US");
	const unsigned max_latency_us = lat ? (unsigned)std::atoi(lat) : 0u;
	struct Queued {
		Op op;
		uint64_t seq;
	};
	std::vector<std::vector<Queued>> queues;  // one per (peer, direction), heads first
	std::vector<int> key;
	for (const Op &op : ops) {
		const int k = op.peer * 2 + (op.send ? 1 : 0);
		size_t q = 0;
		while (q < key.size() && key[q] != k) q++;
		if (q == key.size()) {
			key.push_back(k);
			queues.emplace_back();
		}
		ncclComm *c = op.comm;
		if (op.send && op.bytes > c->slot) return failed("message larger than a slot (CRD_STANDIN_SLOT_MB)");
		queues[q].push_back(Queued{op, op.send ? c->n_sent[op.peer]++ : c->n_received[op.peer]++});  // (sequence numbers in ISSUE order)
	}
	std::vector<size_t> head(queues.size(), 0);
	size_t left = ops.size();
	const auto t0 = std::chrono::steady_clock::now();
	const double limit = timeout_s();
	unsigned idle = 0;
	while (left > 0) {
		const size_t q = (size_t)(rnd() % queues.size());
		if (head[q] >= queues[q].size()) continue;
		const Queued &it = queues[q][head[q]];
		ncclComm *c = it.op.comm;
		Channel &ch = it.op.send ? c->sh->channel[c->rank][it.op.peer] : c->sh->channel[it.op.peer][c->rank];
		const bool ready = it.op.send ? ch.consumed.load(std::memory_order_acquire) + kSlots > it.seq : ch.sent.load(std::memory_order_acquire) > it.seq;
		if (!ready) {
			if (++idle > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
			if ((idle & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
				return failed(it.op.send ? "timeout: the peer does not receive (send side, shuffled delivery)" : "timeout: the peer does not send (receive side, shuffled delivery)");
			continue;
		}
		idle = 0;
		if (max_latency_us) std::this_thread::sleep_for(std::chrono::microseconds(rnd() % (max_latency_us + 1)));
		if (it.op.send) {
			if (hipMemcpyAsync(c->slot_of(c->rank, it.op.peer, it.seq), it.op.buf, it.op.bytes, hipMemcpyDeviceToHost, it.op.stream) != hipSuccess ||
			    hipStreamSynchronize(it.op.stream) != hipSuccess)
				return failed("device-to-host copy failed");
			ch.bytes[it.seq % kSlots] = it.op.bytes;
			ch.sent.store(it.seq + 1, std::memory_order_release);
		} else {
			if (ch.bytes[it.seq % kSlots] != it.op.bytes) return failed("a receive met a send of another size: the ranks' operations are not paired");
			if (hipMemcpyAsync(it.op.buf, c->slot_of(it.op.peer, c->rank, it.seq), it.op.bytes, hipMemcpyHostToDevice, it.op.stream) != hipSuccess ||
			    hipStreamSynchronize(it.op.stream) != hipSuccess)
				return failed("host-to-device copy failed");
			ch.consumed.store(it.seq + 1, std::memory_order_release);
		}
		head[q]++;
		left--;
	}
	return ncclSuccess;
}

ncclResult_t run_group(std::vector<Op> &ops)
{
	// everything already on the streams first: the data the sends read has been produced
	for (const Op &op : ops)
		if (hipStreamSynchronize(op.stream) != hipSuccess) return failed("hipStreamSynchronize failed");
	if (const char *e = std::getenv("CRD_STANDIN_SHUFFLE")) return run_group_shuffled(ops, (unsigned)std::atoi(e));
	for (const Op &op : ops) {
		if (!op.send) continue;
		ncclComm *c = op.comm;
		Channel &ch = c->sh->channel[c->rank][op.peer];
		const uint64_t seq = c->n_sent[op.peer]++;
		if (op.bytes > c->slot) return failed("message larger than a slot (CRD_STANDIN_SLOT_MB)");
		if (!wait_until([&] { return ch.consumed.load(std::memory_order_acquire) + kSlots > seq; })) return failed("timeout: the peer does not receive (send side)");
		if (hipMemcpyAsync(c->slot_of(c->rank, op.peer, seq), op.buf, op.bytes, hipMemcpyDeviceToHost, op.stream) != hipSuccess ||
		    hipStreamSynchronize(op.stream) != hipSuccess)
			return failed("device-to-host copy failed");
		ch.bytes[seq % kSlots] = op.bytes;
		ch.sent.store(seq + 1, std::memory_order_release);
	}
	for (const Op &op : ops) {
		if (op.send) continue;
		ncclComm *c = op.comm;
		Channel &ch = c->sh->channel[op.peer][c->rank];
		const uint64_t seq = c->n_received[op.peer]++;
		if (!wait_until([&] { return ch.sent.load(std::memory_order_acquire) > seq; })) return failed("timeout: the peer does not send (receive side)");
		if (ch.bytes[seq % kSlots] != op.bytes) return failed("a receive met a send of another size: the ranks' operations are not paired");
		if (hipMemcpyAsync(op.buf, c->slot_of(op.peer, c->rank, seq), op.bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess ||
		    hipStreamSynchronize(op.stream) != hipSuccess)
			return failed("host-to-device copy failed");
		ch.consumed.store(seq + 1, std::memory_order_release);
	}
	return ncclSuccess;
}

ncclResult_t point_to_point(bool send, void *buf, size_t count, ncclDataType_t type, int peer, ncclComm *comm, hipStream_t stream)
{
	if (!comm || peer < 0 || peer >= comm->n || !size_of(type)) return failed("bad argument");
	Op op{send, buf, count * size_of(type), peer, comm, stream};
	if (g_group_depth > 0) {
		g_group_ops.push_back(op);
		return ncclSuccess;
	}
	std::vector<Op> one{op};
	return run_group(one);
}

template <typename T>
void reduce(T *out, const T *in, size_t n, ncclRedOp_t op, bool first)
{
	for (size_t i = 0; i < n; i++) {
		if (first) out[i] = in[i];
		else if (op == ncclSum) out[i] += in[i];
		else if (op == ncclMin) out[i] = in[i] < out[i] ? in[i] : out[i];
		else if (op == ncclMax) out[i] = in[i] > out[i] ? in[i] : out[i];
		else if (op == ncclProd) out[i] *= in[i];
	}
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	std::memset(id, 0, sizeof *id);
	timespec ts;
	clock_gettime(CLOCK_REALTIME, &ts);
	std::snprintf(id->internal, sizeof id->internal, "/crd_ring_standin_%d_%lld", (int)getpid(), (long long)ts.tv_nsec + 1000000000ll * (long long)(ts.tv_sec % 1000));
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
	if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.internal[0] != '/') return failed("bad argument to ncclCommInitRank");
	if (std::getenv("CRD_STANDIN_FAIL_INIT")) return failed("ncclCommInitRank fails on request (CRD_STANDIN_FAIL_INIT)");  // a ring that does not come up
	ncclComm *c = new ncclComm;
	c->rank = rank;
	c->n = nranks;
	c->slot = slot_bytes();
	std::snprintf(c->name, sizeof c->name, "%s", id.internal);
	const size_t header = (sizeof(Shared) + 4095) & ~(size_t)4095;
	c->map_bytes = header + (size_t)kMaxRanks * kMaxRanks * kSlots * c->slot;  // sparse: only the slots of ring neighbours are ever touched
	const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
		delete c;
		return failed("shm_open / ftruncate failed");
	}
	void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
	close(fd);
	if (p == MAP_FAILED) {
		delete c;
		return failed("mmap failed");
	}
	c->sh = static_cast<Shared *>(p);  // (a new segment is zero-filled: every counter starts at 0)
	c->slots = static_cast<unsigned char *>(p) + header;
	c->sh->joined.fetch_add(1);
	if (!wait_until([&] { return c->sh->joined.load() >= nranks; })) {
		shm_unlink(c->name);
		return failed("timeout: not every rank called ncclCommInitRank");
	}
	*comm = c;
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	if (!comm) return ncclSuccess;
	shm_unlink(comm->name);  // (the first rank to leave removes the name; the mappings live on until unmapped)
	munmap(comm->sh, comm->map_bytes);
	delete comm;
	return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
	g_group_depth++;
	return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
	if (g_group_depth <= 0) return failed("ncclGroupEnd without ncclGroupStart");
	if (--g_group_depth > 0) return ncclSuccess;
	std::vector<Op> ops;
	ops.swap(g_group_ops);
	return run_group(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	return point_to_point(true, const_cast<void *>(buf), count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	return point_to_point(false, buf, count, type, peer, comm, stream);
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
	const size_t bytes = count * size_of(type);
	if (!comm || !bytes || bytes > kReduceBytes || (type != ncclDouble && type != ncclFloat)) return failed("all-reduce: unsupported type or size");
	Shared *sh = comm->sh;
	const uint64_t seq = comm->n_reduced++;
	// Slot seq % 2 is free: this rank finished all-reduce seq - 1, so every rank had published seq - 1, which each does only
	// after it has finished reading the slots of seq - 2.
	unsigned char *mine = sh->reduce_data[seq % 2][comm->rank];
	if (hipMemcpyAsync(mine, sendbuf, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
		return failed("device-to-host copy failed");
	sh->reduce_seq[comm->rank].store(seq + 1, std::memory_order_release);
	for (int r = 0; r < comm->n; r++)
		if (!wait_until([&] { return sh->reduce_seq[r].load(std::memory_order_acquire) > seq; })) return failed("timeout: a rank did not join the all-reduce");
	alignas(8) unsigned char result[kReduceBytes];
	for (int r = 0; r < comm->n; r++) {
		const unsigned char *in = sh->reduce_data[seq % 2][r];
		if (type == ncclDouble) reduce(reinterpret_cast<double *>(result), reinterpret_cast<const double *>(in), count, op, r == 0);
		else reduce(reinterpret_cast<float *>(result), reinterpret_cast<const float *>(in), count, op, r == 0);
	}
	if (hipMemcpyAsync(recvbuf, result, bytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
		return failed("host-to-device copy failed");
	return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
	*count = comm->n;
	return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
	*rank = comm->rank;
	return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (g_last_error[0] ? g_last_error : "ring stand-in error"); }

}  // extern "C"
