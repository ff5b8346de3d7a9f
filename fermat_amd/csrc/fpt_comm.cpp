// fpt_comm.cpp — the multi-GPU entry points of the C-ABI ("fpt_gather_framebuffer(ctx, nccl comm, root)"): one process
// per GPU, image tiles sharded over the ranks with no data-path collective, ONE exchange per output image: the tile-owned frame-buffer
// pixels travel to the root over RCCL (xGMI) as grouped ncclSend / ncclRecv on the context's stream; the BPT adds one integer
// all-reduce of its light-tracing splat sums.  This is what a multi-GPU Fermat host would call where the single-GPU reference does
// cudaSetDevice(0) and owns the whole frame (src/renderer.cu:600-603).
//
// RCCL is loaded at run time (dlopen), never linked: single-GPU users of the library do not need it, and a host process that already
// carries a copy (PyTorch ships its own librccl.so.1) gets that same copy instead of a second one.
#include "fpt_host.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>

using namespace fpt;

namespace {

struct RcclApi
{
	void* handle = nullptr;
	decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
	decltype(&ncclCommInitRank)   CommInitRank = nullptr;
	decltype(&ncclCommDestroy)    CommDestroy = nullptr;
	decltype(&ncclSend)           Send = nullptr;
	decltype(&ncclRecv)           Recv = nullptr;
	decltype(&ncclGroupStart)     GroupStart = nullptr;
	decltype(&ncclGroupEnd)       GroupEnd = nullptr;
	decltype(&ncclAllReduce)      AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclCommCount)      CommCount = nullptr;
	decltype(&ncclCommUserRank)   CommUserRank = nullptr;
};

RcclApi& rccl()
{
	static RcclApi api;
	if (api.handle) return api;
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	void* h = nullptr;
	for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;        // a copy the process already holds
	if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
	if (!h) throw std::runtime_error(std::string("RCCL is not available (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?"));
	auto sym = [&](const char* s) { void* p = dlsym(h, s); if (!p) throw std::runtime_error(std::string("RCCL symbol missing: ") + s); return p; };
	api.GetUniqueId    = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
	api.CommInitRank   = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
	api.CommDestroy    = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
	api.Send           = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
	api.Recv           = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
	api.GroupStart     = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
	api.GroupEnd       = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
	api.AllReduce      = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
	api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
	api.CommCount      = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
	api.CommUserRank   = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
	api.handle = h;
	return api;
}

void nccl_check(ncclResult_t r, const char* what)
{
	if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + rccl().GetErrorString(r));
}

thread_local std::string g_comm_error;

// the device copies of the ranks' pixel lists (fpt_set_tile_lists) are a function of the tile rule and the world size
void drop_list_cache(fpt_context* ctx) { ctx->comm_lists.clear(); ctx->tile_counts.clear(); ctx->tile_samples.clear(); ctx->tile_world = 0; ctx->tile_rank = 0; }

// what a caller-passed table is recognised by (fpt_gather_framebuffer): per rank, its count and LIST_SAMPLES evenly spaced entries.  Computed by
// fpt_set_tile_lists itself, so the registered tables and their fingerprint cannot drift apart whoever registered them
static constexpr size_t LIST_SAMPLES = 16;
static std::vector<uint32_t> list_samples(int world_size, const uint32_t* const* h_pixel_lists, const uint32_t* h_counts)
{
	std::vector<uint32_t> samples(size_t(world_size) * LIST_SAMPLES, 0u);
	for (int r = 0; r < world_size; ++r)
	{
		const size_t n = h_counts[r]; const uint32_t* l = h_pixel_lists[r];
		for (size_t k = 0; k < LIST_SAMPLES && n; ++k) samples[LIST_SAMPLES * size_t(r) + k] = l[(k * (n - 1)) / (LIST_SAMPLES - 1)];
	}
	return samples;
}

} // namespace

extern "C" {

int fpt_comm_unique_id(char* out_id /*[128]*/)
{
	try
	{
		static_assert(sizeof(ncclUniqueId) == FPT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
		ncclUniqueId id;
		nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
		std::memcpy(out_id, &id, sizeof(id));
		return 0;
	}
	catch (const std::exception& e) { g_comm_error = e.what(); return 1; }
}
const char* fpt_comm_last_error(void) { return g_comm_error.c_str(); }

int fpt_comm_init(fpt_context* ctx, int rank, int world_size, const char* id /*[128]*/)
{
	return guarded(ctx, [&] {
		require(world_size >= 1 && rank >= 0 && rank < world_size && id, "fpt_comm_init: bad rank / world size / id");
		require(ctx->comm == nullptr, "fpt_comm_init: this context already has a communicator");
		ncclUniqueId uid; std::memcpy(&uid, id, sizeof(uid));
		ncclComm_t c = nullptr;
		nccl_check(rccl().CommInitRank(&c, world_size, uid, rank), "ncclCommInitRank");
		ctx->comm = c; ctx->comm_rank = rank; ctx->comm_world = world_size; ctx->comm_owned = true;
		drop_list_cache(ctx);
	});
}
int fpt_comm_adopt(fpt_context* ctx, void* nccl_comm, int rank, int world_size)
{
	return guarded(ctx, [&] {
		require(nccl_comm && world_size >= 1 && rank >= 0 && rank < world_size, "fpt_comm_adopt: bad communicator / rank / world size");
		require(ctx->comm == nullptr, "fpt_comm_adopt: this context already has a communicator (fpt_comm_destroy first)");
		(void)rccl();
		ctx->comm = nccl_comm; ctx->comm_rank = rank; ctx->comm_world = world_size; ctx->comm_owned = false;
		drop_list_cache(ctx);
	});
}
int fpt_comm_destroy(fpt_context* ctx)
{
	return guarded(ctx, [&] {
		if (ctx->comm && ctx->comm_owned) { FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream)); nccl_check(rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm)), "ncclCommDestroy"); }
		ctx->comm = nullptr; ctx->comm_world = 1; ctx->comm_rank = 0; ctx->comm_owned = false;
		drop_list_cache(ctx);
	});
}
// what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): lets a host record that the collective really spans N ranks
int fpt_comm_info(fpt_context* ctx, int* rank, int* world_size)
{
	return guarded(ctx, [&] {
		require(ctx->comm != nullptr, "fpt_comm_info: no communicator (fpt_comm_init / fpt_comm_adopt)");
		int n = 0, r = 0;
		nccl_check(rccl().CommCount(static_cast<ncclComm_t>(ctx->comm), &n), "ncclCommCount");
		nccl_check(rccl().CommUserRank(static_cast<ncclComm_t>(ctx->comm), &r), "ncclCommUserRank");
		if (rank) *rank = r;
		if (world_size) *world_size = n;
	});
}

// ---- the tile tables and the two halves of the gather -----------------------------------------------------------------------------
// Rank r owns the pixels h_pixel_lists[r][0 .. h_counts[r]) (absolute indices; every rank passes the same tables -- they are a pure function of the
// tile rule).  fpt_set_tile_lists uploads what this rank needs ONCE (its own list; rank `root`: everybody's): until round 3 every gather call hashed all
// lists on the host to see whether they had changed -- 1.5 ms per call at 1600x900, inside the timed region.
int fpt_set_tile_lists(fpt_context* ctx, int rank, int world_size, int root, const uint32_t* const* h_pixel_lists, const uint32_t* h_counts)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(world_size >= 1 && rank >= 0 && rank < world_size && root >= 0 && root < world_size, "fpt_set_tile_lists: bad rank / world size / root");
		require(h_pixel_lists && h_counts, "fpt_set_tile_lists: null argument");
		require(ctx->comm == nullptr || (ctx->comm_world == world_size && ctx->comm_rank == rank), "fpt_set_tile_lists: rank / world size differ from the communicator's");
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		ctx->comm_lists.clear(); ctx->comm_lists.resize(size_t(world_size));
		for (int r = 0; r < world_size; ++r)
		{
			ctx->comm_lists[size_t(r)].reset(new DeviceArray<uint32_t>());
			if ((r == rank || rank == root) && h_counts[r]) ctx->comm_lists[size_t(r)]->upload(h_pixel_lists[r], h_counts[r], ctx->stream);
		}
		ctx->tile_counts.assign(h_counts, h_counts + world_size);
		ctx->tile_samples = list_samples(world_size, h_pixel_lists, h_counts);
		ctx->tile_world = world_size; ctx->tile_rank = rank; ctx->tile_root = root;
	});
}

static std::vector<int> gather_channels(const fpt_rendering_context_view* view, uint32_t channel_mask, const char* who)
{
	std::vector<int> channels;
	for (int c = 0; c < FPT_FB_NUM_CHANNELS; ++c)
		if (channel_mask & (1u << c)) { require(view->fb.channels[c] != nullptr, who); channels.push_back(c); }
	return channels;
}

// pack half: this rank's owned pixels of every requested channel -> ONE contiguous message in the library's staging buffer (channel-major: all pixels of the
// first requested channel, then the next), on the context's stream.  *d_message stays valid until the next pack / gather of this context.
int fpt_gather_pack(fpt_context* ctx, const fpt_rendering_context_view* view, uint32_t channel_mask, const float** d_message, uint64_t* n_floats)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->tile_world > 0, "fpt_gather_pack: fpt_set_tile_lists has not been called");
		require(view != nullptr, "fpt_gather_pack: null view");
		const std::vector<int> channels = gather_channels(view, channel_mask, "fpt_gather_pack: null channel");
		const uint32_t n = ctx->tile_counts[size_t(ctx->tile_rank)];
		ctx->comm_staging.alloc(std::max<size_t>(size_t(n) * channels.size(), 1));
		for (size_t k = 0; k < channels.size() && n; ++k)
			launch_pack_pixels(reinterpret_cast<const float4*>(view->fb.channels[channels[k]]), ctx->comm_lists[size_t(ctx->tile_rank)]->ptr, n, ctx->comm_staging.ptr + size_t(n) * k, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
		if (d_message) *d_message = reinterpret_cast<const float*>(ctx->comm_staging.ptr);
		if (n_floats) *n_floats = uint64_t(n) * channels.size() * 4u;
	});
}
// unpack half (on the rank the tile tables were registered with as root): a message rank `src_rank` packed -> that rank's pixels of this context's frame
// buffer, on the context's stream.  d_message is device memory of THIS context's device (the host moved it: RCCL, hipMemcpyPeer, a file ...)
int fpt_gather_unpack(fpt_context* ctx, const fpt_rendering_context_view* view, uint32_t channel_mask, int src_rank, const float* d_message)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->tile_world > 0 && ctx->tile_rank == ctx->tile_root, "fpt_gather_unpack: the tile tables are not registered on this context as the root's (fpt_set_tile_lists)");
		require(view && d_message && src_rank >= 0 && src_rank < ctx->tile_world, "fpt_gather_unpack: bad argument");
		const std::vector<int> channels = gather_channels(view, channel_mask, "fpt_gather_unpack: null channel");
		const uint32_t n = ctx->tile_counts[size_t(src_rank)];
		for (size_t k = 0; k < channels.size() && n; ++k)
			launch_unpack_pixels(reinterpret_cast<const float4*>(d_message) + size_t(n) * k, ctx->comm_lists[size_t(src_rank)]->ptr, n, reinterpret_cast<float4*>(view->fb.channels[channels[k]]), ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}

// Gather = pack on every other rank, ONE RCCL group of sends / receives on the context's stream, unpack on the root; nothing synchronises the host and
// nothing is hashed or uploaded per call.  h_pixel_lists / h_counts may be NULL once fpt_set_tile_lists has registered the tables; passing them registers
// them when they are not the registered ones (compared by rank count, counts and 16 evenly spaced entries per list, the fingerprint fpt_set_tile_lists itself
// records -- a list edited in place between those entries must be re-registered with fpt_set_tile_lists).
int fpt_gather_framebuffer(fpt_context* ctx, const fpt_rendering_context_view* view, int root, uint32_t channel_mask,
                           const uint32_t* const* h_pixel_lists, const uint32_t* h_counts)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->comm != nullptr, "fpt_gather_framebuffer: no communicator (fpt_comm_init / fpt_comm_adopt)");
		require(view != nullptr, "fpt_gather_framebuffer: null view");
		const int W = ctx->comm_world, me = ctx->comm_rank;
		require(root >= 0 && root < W, "fpt_gather_framebuffer: bad root");
		if (h_pixel_lists && h_counts)
		{
			bool same = ctx->tile_world == W && ctx->tile_rank == me && ctx->tile_root == root && ctx->tile_samples.size() == size_t(W) * LIST_SAMPLES;
			for (int r = 0; r < W && same; ++r) same = ctx->tile_counts[size_t(r)] == h_counts[r];
			same = same && list_samples(W, h_pixel_lists, h_counts) == ctx->tile_samples;
			if (!same) require(fpt_set_tile_lists(ctx, me, W, root, h_pixel_lists, h_counts) == 0, ctx->error.c_str());
		}
		require(ctx->tile_world == W && ctx->tile_rank == me && ctx->tile_root == root, "fpt_gather_framebuffer: no tile tables for this communicator and root (fpt_set_tile_lists)");
		const std::vector<int> channels = gather_channels(view, channel_mask, "fpt_gather_framebuffer: null channel");
		if (channels.empty()) return;
		hipStream_t s = ctx->stream;
		ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
		const size_t n_ch = channels.size();
		if (me != root)
		{
			const float* msg = nullptr; uint64_t n_floats = 0;
			require(fpt_gather_pack(ctx, view, channel_mask, &msg, &n_floats) == 0, ctx->error.c_str());
			if (n_floats == 0) return;
			nccl_check(rccl().GroupStart(), "ncclGroupStart");
			nccl_check(rccl().Send(msg, size_t(n_floats), ncclFloat, root, comm, s), "ncclSend");
			nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
			return;
		}
		size_t total = 0;
		std::vector<size_t> offset(size_t(W), 0);
		for (int r = 0; r < W; ++r) if (r != root) { offset[size_t(r)] = total; total += size_t(ctx->tile_counts[size_t(r)]) * n_ch; }
		ctx->comm_recv.alloc(std::max<size_t>(total, 1));
		nccl_check(rccl().GroupStart(), "ncclGroupStart");
		for (int r = 0; r < W; ++r)
			if (r != root && ctx->tile_counts[size_t(r)])
				nccl_check(rccl().Recv(ctx->comm_recv.ptr + offset[size_t(r)], size_t(ctx->tile_counts[size_t(r)]) * n_ch * 4, ncclFloat, r, comm, s), "ncclRecv");
		nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
		for (int r = 0; r < W; ++r)
			if (r != root && ctx->tile_counts[size_t(r)])
				require(fpt_gather_unpack(ctx, view, channel_mask, r, reinterpret_cast<const float*>(ctx->comm_recv.ptr + offset[size_t(r)])) == 0, ctx->error.c_str());
	});
}

// BPT under tile sharding: the light-tracing splat sums (3 x int64 per pixel per pass in flight, 2^-32 fixed point, order-independent) are
// summed over the ranks in place; every rank then calls fpt_bpt_resolve_splats
int fpt_bpt_allreduce_splats(fpt_context* ctx, uint64_t n_int64)
{
	return guarded(ctx, [&] {
		require(ctx->comm != nullptr, "fpt_bpt_allreduce_splats: no communicator (fpt_comm_init / fpt_comm_adopt)");
		long long* p = ctx->bpt.splat_ptr();
		require(p != nullptr && n_int64 > 0, "fpt_bpt_allreduce_splats: no splat buffer");
		nccl_check(rccl().AllReduce(p, p, size_t(n_int64), ncclInt64, ncclSum, static_cast<ncclComm_t>(ctx->comm), ctx->stream), "ncclAllReduce");
	});
}

// PSFPT under tile sharding: every rank hands every other rank the cells its paths touched in the pass in flight (PsfRecord: key, three
// fixed-point sums, count -- a few thousand per pass on the bench frame, 40 B each) and merges all of them, its own included, into its copy
// of the global table; fpt_psfpt_finish then blends from equal tables on every rank.  One integer all-reduce tells everybody the counts, one
// RCCL group carries the records.
int fpt_psfpt_exchange_cells(fpt_context* ctx)
{
	return guarded(ctx, [&] {
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.sharded && ps.pending, "fpt_psfpt_exchange_cells: no sharded pass is pending (fpt_psfpt_set_sharded, fpt_psfpt_render)");
		require(ctx->comm != nullptr, "fpt_psfpt_exchange_cells: no communicator (fpt_comm_init / fpt_comm_adopt)");
		const int W = ctx->comm_world, me = ctx->comm_rank;
		ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
		hipStream_t s = ctx->stream;
		ps.ex_counts.alloc(std::max<size_t>(ps.ex_counts.count, size_t(W)));
		FPT_HIP_CHECK(hipMemsetAsync(ps.ex_counts.ptr, 0, size_t(W) * sizeof(uint32_t), s));
		FPT_HIP_CHECK(hipMemcpyAsync(ps.ex_counts.ptr + me, ps.touched_n.ptr, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
		if (W > 1) nccl_check(rccl().AllReduce(ps.ex_counts.ptr, ps.ex_counts.ptr, size_t(W), ncclUint32, ncclSum, comm, s), "ncclAllReduce");
		std::vector<uint32_t> counts(size_t(W), 0u);
		ps.ex_counts.download(counts.data(), size_t(W), s);
		size_t others = 0;
		for (int r = 0; r < W; ++r) if (r != me) others += counts[r];
		ps.recv.alloc(std::max<size_t>(ps.recv.count, std::max<size_t>(others, 1)));
		if (W > 1)
		{
			nccl_check(rccl().GroupStart(), "ncclGroupStart");
			size_t off = 0;
			for (int r = 0; r < W; ++r)
			{
				if (r == me) continue;
				if (counts[me]) nccl_check(rccl().Send(ps.records.ptr, size_t(counts[me]) * sizeof(PsfRecord), ncclChar, r, comm, s), "ncclSend");
				if (counts[r])  nccl_check(rccl().Recv(ps.recv.ptr + off, size_t(counts[r]) * sizeof(PsfRecord), ncclChar, r, comm, s), "ncclRecv");
				off += counts[r];
			}
			nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
		}
		// integer sums: the order of the merges does not matter
		if (counts[me]) require(fpt_psfpt_import_cells(ctx, ps.records.ptr, counts[me]) == 0, "fpt_psfpt_exchange_cells: merging this rank's cells failed");
		if (others)     require(fpt_psfpt_import_cells(ctx, ps.recv.ptr, uint32_t(others)) == 0, "fpt_psfpt_exchange_cells: merging the other ranks' cells failed");
	});
}

// BPT -sc 1 under tile sharding, the same image for any number of ranks: every rank hands every other rank the light vertices its light paths stored
// in the batch in flight (80-byte records: store slot + vertex), so that fpt_bpt_finish builds the list of ALL light vertices everywhere.  One integer
// all-reduce tells everybody the counts, one RCCL group carries the records (an all-gather of unequal parts).
int fpt_bpt_exchange_light_vertices(fpt_context* ctx)
{
	return guarded(ctx, [&] {
		fpt_context::BptState& b = ctx->bpt;
		require(ctx->comm != nullptr, "fpt_bpt_exchange_light_vertices: no communicator (fpt_comm_init / fpt_comm_adopt)");
		const int W = ctx->comm_world, me = ctx->comm_rank;
		ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
		hipStream_t s = ctx->stream;
		const uint32_t mine = bpt_pack_own_vertices(ctx);
		DeviceArray<uint32_t>& dc = ctx->psf.ex_counts;                   // a few words of scratch shared with the PSFPT exchange
		dc.alloc(std::max<size_t>(dc.count, size_t(W)));
		std::vector<uint32_t> counts(size_t(W), 0u); counts[size_t(me)] = mine;
		FPT_HIP_CHECK(hipMemcpyAsync(dc.ptr, counts.data(), size_t(W) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
		if (W > 1) nccl_check(rccl().AllReduce(dc.ptr, dc.ptr, size_t(W), ncclUint32, ncclSum, comm, s), "ncclAllReduce");
		dc.download(counts.data(), size_t(W), s);
		size_t others = 0;
		for (int r = 0; r < W; ++r) if (r != me) others += counts[size_t(r)];
		b.lv_recv.alloc(std::max<size_t>(b.lv_recv.count, std::max<size_t>(others, 1)));
		if (W > 1)
		{
			nccl_check(rccl().GroupStart(), "ncclGroupStart");
			size_t off = 0;
			for (int r = 0; r < W; ++r)
			{
				if (r == me) continue;
				if (mine) nccl_check(rccl().Send(b.lv_send.ptr, size_t(mine) * sizeof(LightVertexWire), ncclChar, r, comm, s), "ncclSend");
				if (counts[size_t(r)]) nccl_check(rccl().Recv(b.lv_recv.ptr + off, size_t(counts[size_t(r)]) * sizeof(LightVertexWire), ncclChar, r, comm, s), "ncclRecv");
				off += counts[size_t(r)];
			}
			nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
		}
		if (others) bpt_import_vertices(ctx, b.lv_recv.ptr, uint32_t(others));
	});
}

// exercise the whole RCCL path on ONE rank (a 1-rank communicator sending a message to itself inside a group): dlopen, the symbols,
// communicator set-up and the stream ordering can be checked on a single-GPU box
int fpt_comm_selftest(fpt_context* ctx, uint32_t n_floats)
{
	return guarded(ctx, [&] {
		require(ctx->comm != nullptr && ctx->comm_world == 1, "fpt_comm_selftest: needs a 1-rank communicator");
		DeviceArray<float> a, b;
		std::vector<float> h(n_floats), back(n_floats, 0.0f);
		for (uint32_t i = 0; i < n_floats; ++i) h[i] = float(i) * 0.5f - 3.0f;
		a.upload(h.data(), n_floats, ctx->stream); b.alloc(n_floats);
		FPT_HIP_CHECK(hipMemsetAsync(b.ptr, 0, n_floats * sizeof(float), ctx->stream));
		ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
		nccl_check(rccl().GroupStart(), "ncclGroupStart");
		nccl_check(rccl().Send(a.ptr, n_floats, ncclFloat, 0, comm, ctx->stream), "ncclSend");
		nccl_check(rccl().Recv(b.ptr, n_floats, ncclFloat, 0, comm, ctx->stream), "ncclRecv");
		nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
		b.download(back.data(), n_floats, ctx->stream);
		require(std::memcmp(h.data(), back.data(), n_floats * sizeof(float)) == 0, "fpt_comm_selftest: the message did not arrive intact");
	});
}

} // extern "C"
