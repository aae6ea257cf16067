// r1_multi.cpp — one process, N GPUs: the frame's tiles split over N devices (tile t -> device
// t % N) and ONE RCCL all-gather of the per-device records (dense tile block + 8-byte ray count)
// at the end of the frame; the row-major image is assembled on device 0 and copied to the host
// once.  This is the multi-GPU form of benchmark()'s join (rayweek1.cpp:804-813: the threads'
// `out_num_rays` are summed and every tile has been written to the one pixel buffer,
// rayweek1.cpp:773-775) that keeps the reference's four-argument benchmark() a single call:
// SURVEY.md §8e "single-process-multi-GPU (ncclCommInitAll) keeps the benchmark() signature
// intact".  bench.py's N-process form (torch.distributed) uses the same record layout.
//
// RCCL is bound at run time (dlopen of librccl.so.1): librays1.so keeps loading where RCCL is
// not installed, and inside a process that already carries a copy (torch bundles one) the same
// library is used instead of a second one.  There is no fallback: r1_multi_create fails if RCCL
// or the devices are not there.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/rays1.h"

extern "C" void r1_set_error(const char *fmt, ...);
extern "C" void *r1_context_stream(r1_context *c); // r1_capi.cpp (internal): the context's own stream

namespace
{

// the slice of rccl.h this file needs (opaque communicator, result code 0 = success)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
enum
{
    R1_NCCL_UINT8 = 1 // ncclUint8 / ncclChar's unsigned twin in rccl.h's ncclDataType_t
};

struct Rccl
{
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};

bool load_rccl(Rccl &r)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!r.lib)
    {
        r1_set_error("r1_multi_create: cannot load librccl.so.1 (%s)", dlerror());
        return false;
    }
#define R1_SYM(field, name)                                                                                            \
    if (!(*(void **)(&r.field) = dlsym(r.lib, name)))                                                                  \
    {                                                                                                                  \
        r1_set_error("r1_multi_create: librccl has no %s", name);                                                      \
        return false;                                                                                                  \
    }
    R1_SYM(CommInitAll, "ncclCommInitAll")
    R1_SYM(CommDestroy, "ncclCommDestroy")
    R1_SYM(AllGather, "ncclAllGather")
    R1_SYM(GroupStart, "ncclGroupStart")
    R1_SYM(GroupEnd, "ncclGroupEnd")
    R1_SYM(GetErrorString, "ncclGetErrorString")
    R1_SYM(GetVersion, "ncclGetVersion")
#undef R1_SYM
    return true;
}

} // namespace

struct r1_multi
{
    Rccl rccl;
    int n = 0;
    std::vector<int> device;
    std::vector<r1_context *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<void *> d_record, d_gathered; // per device: its record, all records
    void *d_rgb = nullptr;                      // device 0: the assembled image
    size_t record_cap = 0, rgb_cap = 0;
    std::vector<hipEvent_t> ev0, ev1;
    int rccl_version = 0;
};

#define R1M_HIP(call)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            r1_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);                   \
            return e_ == hipErrorOutOfMemory ? R1_ENOMEM : R1_EHIP;                                                    \
        }                                                                                                              \
    } while (0)
#define R1M_NCCL(m, call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        ncclResult_t r_ = (call);                                                                                      \
        if (r_ != 0)                                                                                                   \
        {                                                                                                              \
            r1_set_error("%s failed: %s (%s:%d)", #call, (m)->rccl.GetErrorString(r_), __FILE__, __LINE__);            \
            return R1_EHIP;                                                                                            \
        }                                                                                                              \
    } while (0)

extern "C" void r1_multi_destroy(r1_multi *m)
{
    if (!m)
        return;
    for (int i = 0; i < (int)m->ctx.size(); ++i)
    {
        (void)hipSetDevice(m->device[i]);
        if (i < (int)m->stream.size() && m->stream[i])
            (void)hipStreamSynchronize(m->stream[i]);
        if (i < (int)m->comm.size() && m->comm[i])
            (void)m->rccl.CommDestroy(m->comm[i]);
        if (i < (int)m->d_record.size() && m->d_record[i])
            (void)hipFree(m->d_record[i]);
        if (i < (int)m->d_gathered.size() && m->d_gathered[i])
            (void)hipFree(m->d_gathered[i]);
        if (i == 0 && m->d_rgb)
            (void)hipFree(m->d_rgb);
        if (i < (int)m->ev0.size() && m->ev0[i])
            (void)hipEventDestroy(m->ev0[i]);
        if (i < (int)m->ev1.size() && m->ev1[i])
            (void)hipEventDestroy(m->ev1[i]);
        // (m->stream[i] is the context's own stream: destroyed with the context)
        r1_destroy(m->ctx[i]);
    }
    // the library handle stays open: other communicators of the process may live in it
    delete m;
}

extern "C" int r1_multi_create(int32_t n_devices, const int32_t *devices, r1_multi **out)
{
    if (!out || n_devices < 1 || n_devices > 64)
    {
        r1_set_error("r1_multi_create: bad argument");
        return R1_EINVAL;
    }
    *out = nullptr;
    const int visible = r1_device_count();
    if (visible <= 0)
        return R1_ENODEVICE;
    r1_multi *m = new (std::nothrow) r1_multi();
    if (!m)
        return R1_ENOMEM;
    m->n = n_devices;
    for (int i = 0; i < n_devices; ++i)
    {
        const int d = devices ? devices[i] : i;
        for (int k = 0; k < i; ++k)
            if (m->device[k] == d)
            {
                r1_set_error("r1_multi_create: device %d listed twice (an RCCL communicator needs distinct devices)", d);
                delete m;
                return R1_EINVAL;
            }
        if (d < 0 || d >= visible)
        {
            r1_set_error("r1_multi_create: device %d out of range (0..%d)", d, visible - 1);
            delete m;
            return R1_EINVAL;
        }
        m->device.push_back(d);
    }
    if (!load_rccl(m->rccl))
    {
        delete m;
        return R1_ENODEVICE;
    }
    (void)m->rccl.GetVersion(&m->rccl_version);
    m->comm.assign(n_devices, nullptr);
    m->stream.assign(n_devices, nullptr);
    m->d_record.assign(n_devices, nullptr);
    m->d_gathered.assign(n_devices, nullptr);
    m->ev0.assign(n_devices, nullptr);
    m->ev1.assign(n_devices, nullptr);
    int rc = R1_OK;
    for (int i = 0; i < n_devices && rc == R1_OK; ++i)
    {
        r1_context *c = nullptr;
        rc = r1_create(m->device[i], &c);
        if (rc == R1_OK)
            m->ctx.push_back(c);
    }
    auto fail = [&](int code) {
        r1_multi_destroy(m);
        return code;
    };
    if (rc != R1_OK)
        return fail(rc);
    for (int i = 0; i < n_devices; ++i)
    {
        m->stream[i] = (hipStream_t)r1_context_stream(m->ctx[i]); // one stream per device: the context's own
        if (hipSetDevice(m->device[i]) != hipSuccess || !m->stream[i] ||
            hipEventCreate(&m->ev0[i]) != hipSuccess || hipEventCreate(&m->ev1[i]) != hipSuccess)
        {
            r1_set_error("r1_multi_create: stream/event creation failed on device %d", m->device[i]);
            return fail(R1_EHIP);
        }
    }
    const ncclResult_t nr = m->rccl.CommInitAll(m->comm.data(), n_devices, m->device.data());
    if (nr != 0)
    {
        r1_set_error("ncclCommInitAll(%d devices) failed: %s", n_devices, m->rccl.GetErrorString(nr));
        return fail(R1_EHIP);
    }
    *out = m;
    return R1_OK;
}

extern "C" int r1_multi_set_scene(r1_multi *m, const r1_scene *scene, const r1_camera *camera)
{
    if (!m)
        return R1_EINVAL;
    for (int i = 0; i < m->n; ++i)
    {
        const int rc = r1_set_scene(m->ctx[i], scene, camera); // replicated: <= 14 KB of tables
        if (rc != R1_OK)
            return rc;
    }
    return R1_OK;
}

// device buffers of one r1_multi: every device's record + gathered records, device 0's image (or frame record)
static int multi_buffers(r1_multi *m, const size_t record, const size_t rgb_bytes)
{
    if (record > m->record_cap)
    {
        for (int i = 0; i < m->n; ++i)
        {
            R1M_HIP(hipSetDevice(m->device[i]));
            R1M_HIP(hipStreamSynchronize(m->stream[i]));
            if (m->d_record[i])
                R1M_HIP(hipFree(m->d_record[i]));
            if (m->d_gathered[i])
                R1M_HIP(hipFree(m->d_gathered[i]));
            m->d_record[i] = m->d_gathered[i] = nullptr;
            R1M_HIP(hipMalloc(&m->d_record[i], record));
            R1M_HIP(hipMalloc(&m->d_gathered[i], record * (size_t)m->n));
        }
        m->record_cap = record;
    }
    if (rgb_bytes > m->rgb_cap)
    {
        R1M_HIP(hipSetDevice(m->device[0]));
        R1M_HIP(hipStreamSynchronize(m->stream[0]));
        if (m->d_rgb)
            R1M_HIP(hipFree(m->d_rgb));
        m->d_rgb = nullptr;
        R1M_HIP(hipMalloc(&m->d_rgb, rgb_bytes));
        m->rgb_cap = rgb_bytes;
    }
    return R1_OK;
}

// Where everything lies in the buffers of an N-device frame (or batch of frames) — pure arithmetic, no device: what every r1_multi_*
// entry point below sizes and addresses its buffers with, exported so that the layout can be checked for any N on a machine with one GPU
// or none (tests/test_multirank_cpu.py compares it with the host mirror of the rank path, rays1bench_amd/sharding.py).
extern "C" int r1_multi_layout(const r1_params *params, int32_t n_devices, int32_t n_frames, r1_multi_layout_info *out)
{
    if (!params || !out || n_devices < 1 || n_frames < 1)
    {
        r1_set_error("r1_multi_layout: bad argument");
        return R1_EINVAL;
    }
    r1_params p = *params;
    p.shard = 0, p.num_shards = n_devices;
    const size_t block = r1_shard_block_bytes(&p), record = r1_shard_record_bytes(&p), frame = r1_frame_record_bytes(&p);
    if (block == 0 || record == 0 || frame == 0)
        return R1_EINVAL;
    memset(out, 0, sizeof(*out));
    out->block_bytes = block;                                  // a device's dense tile block of ONE frame
    out->record_bytes = record;                                // block padded to 8 bytes + the device's uint64 ray count
    out->count_offset = record - 8;                            // ... the count's place in a record
    out->send_bytes = record * (size_t)n_frames;               // what a device hands to the all-gather: its n_frames records
    out->gathered_bytes = out->send_bytes * (size_t)n_devices; // what it receives: [device][frame][record]
    out->frame_record_bytes = frame;                           // an assembled frame: row-major image padded to 8 bytes + the frame's uint64 count
    out->frame_count_offset = frame - 8;
    out->host_bytes = frame * (size_t)n_frames;                // the one copy to the host
    out->counts_pitch = record * (size_t)n_frames;             // single frames: the N counts are a strided 8-byte column of the gathered buffer
    return R1_OK;
}

// record of device d, frame f in a gathered buffer
static inline size_t gathered_at(const r1_multi_layout_info &L, int32_t n_frames, int d, int f) { return ((size_t)d * n_frames + f) * L.record_bytes; }

extern "C" int r1_multi_render(r1_multi *m, const r1_params *params, uint8_t *rgb_out, uint64_t *num_rays_out, double *device_seconds_out)
{
    if (!m || !params || !rgb_out)
    {
        r1_set_error("r1_multi_render: null argument");
        return R1_EINVAL;
    }
    r1_params p = *params;
    p.shard = 0, p.num_shards = m->n;
    r1_multi_layout_info L;
    if (r1_multi_layout(&p, m->n, 1, &L) != R1_OK)
        return R1_EINVAL;
    const size_t record = L.record_bytes; // block, padded to 8 bytes, + the shard's uint64 ray count: pixels and counts travel in one collective
    const size_t trailer = L.count_offset;
    const size_t rgb_bytes = (size_t)p.width * p.height * 3;
    {
        const int rc_buf = multi_buffers(m, record, rgb_bytes);
        if (rc_buf != R1_OK)
            return rc_buf;
    }
    // 1. every device traces + resolves its tiles into its record (latency-mode kernels: one frame, the caller waits)
    // From the first enqueue on, a failure must not leave work in flight behind the caller's back: the error is
    // remembered, an open RCCL group is closed, every device's stream is drained, and only then does the call return.
    int first_rc = R1_OK;
    auto drain = [&](int rc) {
        for (int i = 0; i < m->n; ++i)
            if (hipSetDevice(m->device[i]) == hipSuccess)
                (void)hipStreamSynchronize(m->stream[i]);
        return rc;
    };
    for (int i = 0; i < m->n && first_rc == R1_OK; ++i)
    {
        r1_params q = p;
        q.shard = i;
        hipError_t he = hipSetDevice(m->device[i]);
        if (he == hipSuccess)
            he = hipEventRecord(m->ev0[i], m->stream[i]);
        if (he != hipSuccess)
        {
            r1_set_error("r1_multi_render: device %d: %s", m->device[i], hipGetErrorString(he));
            first_rc = R1_EHIP;
            break;
        }
        first_rc = r1_render_shard_device_once(m->ctx[i], &q, m->d_record[i], (char *)m->d_record[i] + trailer, m->stream[i]);
    }
    if (first_rc != R1_OK)
        return drain(first_rc);
    // 2. the one exchange step of the frame: all-gather of the records over xGMI
    ncclResult_t nr = m->rccl.GroupStart();
    if (nr != 0)
    {
        r1_set_error("ncclGroupStart failed: %s", m->rccl.GetErrorString(nr));
        return drain(R1_EHIP);
    }
    for (int i = 0; i < m->n && nr == 0; ++i)
        nr = m->rccl.AllGather(m->d_record[i], m->d_gathered[i], record, R1_NCCL_UINT8, m->comm[i], m->stream[i]);
    const ncclResult_t ne = m->rccl.GroupEnd(); // always closed, also after a failed enqueue
    if (nr != 0 || ne != 0)
    {
        r1_set_error("ncclAllGather of the frame's records failed: %s", m->rccl.GetErrorString(nr != 0 ? nr : ne));
        return drain(R1_EHIP);
    }
    for (int i = 0; i < m->n; ++i)
    {
        if (hipSetDevice(m->device[i]) != hipSuccess || hipEventRecord(m->ev1[i], m->stream[i]) != hipSuccess)
        {
            r1_set_error("r1_multi_render: event record failed on device %d", m->device[i]);
            return drain(R1_EHIP);
        }
    }
    // 3. device 0 scatters the gathered tiles into the row-major image; one copy to the host (+ the N counts)
    std::vector<uint64_t> counts((size_t)m->n, 0);
    {
        hipError_t he = hipSetDevice(m->device[0]);
        int rc = he == hipSuccess ? r1_assemble_device_strided(m->ctx[0], &p, m->d_gathered[0], record, m->d_rgb, m->stream[0]) : R1_EHIP;
        if (rc == R1_OK)
        {
            he = hipMemcpyAsync(rgb_out, m->d_rgb, rgb_bytes, hipMemcpyDeviceToHost, m->stream[0]);
            if (he == hipSuccess)
                he = hipMemcpy2DAsync(counts.data(), 8, (char *)m->d_gathered[0] + gathered_at(L, 1, 0, 0) + trailer, L.counts_pitch, 8, (size_t)m->n, hipMemcpyDeviceToHost, m->stream[0]);
            if (he != hipSuccess)
            {
                r1_set_error("r1_multi_render: copy to the host failed: %s", hipGetErrorString(he));
                rc = R1_EHIP;
            }
        }
        // every device's stream is idle before the call returns (the other devices' gathers are complete before their
        // records are reused), whether the frame succeeded or not
        for (int i = 0; i < m->n; ++i)
        {
            he = hipSetDevice(m->device[i]);
            if (he == hipSuccess)
                he = hipStreamSynchronize(m->stream[i]);
            if (he != hipSuccess && rc == R1_OK)
            {
                r1_set_error("r1_multi_render: device %d: %s", m->device[i], hipGetErrorString(he));
                rc = R1_EHIP;
            }
        }
        if (rc != R1_OK)
            return rc;
    }
    uint64_t rays = 0;
    for (uint64_t c : counts)
        rays += c; // rayweek1.cpp:809-813
    if (num_rays_out)
        *num_rays_out = rays;
    if (device_seconds_out)
    {
        double worst = 0;
        for (int i = 0; i < m->n; ++i)
        {
            float ms = 0;
            R1M_HIP(hipSetDevice(m->device[i]));
            R1M_HIP(hipEventElapsedTime(&ms, m->ev0[i], m->ev1[i]));
            worst = ms > worst ? ms : worst;
        }
        *device_seconds_out = worst * 1e-3; // render + gather, slowest device
    }
    return R1_OK;
}

// Frames in flight across N GPUs from ONE process: the enqueue-only form of r1_multi_render, for one frame or a BATCH of frames
// (r1_render_shard_device_batch on every device, ONE all-gather of the n_frames records per device, one assemble, one copy: a device's
// share of a single frame is a small launch at N >= 4).  Every device renders its tiles with
// the THROUGHPUT kernels (few long-lived waves per frame), the records go through this r1_multi's own all-gather, device 0 assembles
// image + summed ray count into one frame record and copies it into `host_frame` (r1_frame_record_bytes, page-locked) — all on the
// object's streams, nothing is waited for.  A caller keeps K frames in flight with K r1_multi objects (each has its own communicator:
// collectives of different frames never share one), exactly as K r1_contexts do on one GPU; r1_multi_sync waits for this object's frame.
extern "C" int r1_multi_render_batch_async(r1_multi *m, const r1_params *params, int32_t n_frames, uint32_t seed_stride, void *host_frames)
{
    if (!m || !params || !host_frames || n_frames < 1)
    {
        r1_set_error("r1_multi_render_batch_async: bad argument");
        return R1_EINVAL;
    }
    r1_params p = *params;
    p.shard = 0, p.num_shards = m->n;
    r1_multi_layout_info L;
    if (r1_multi_layout(&p, m->n, n_frames, &L) != R1_OK)
        return R1_EINVAL;
    const size_t record = L.record_bytes, frame = L.frame_record_bytes;
    {
        const int rc_buf = multi_buffers(m, L.send_bytes, L.host_bytes);
        if (rc_buf != R1_OK)
            return rc_buf;
    }
    auto drain = [&](int rc) {
        for (int i = 0; i < m->n; ++i)
            if (hipSetDevice(m->device[i]) == hipSuccess)
                (void)hipStreamSynchronize(m->stream[i]);
        return rc;
    };
    int rc = R1_OK;
    for (int i = 0; i < m->n && rc == R1_OK; ++i)
    {
        r1_params q = p;
        q.shard = i;
        if (hipSetDevice(m->device[i]) != hipSuccess)
        {
            r1_set_error("r1_multi_render_batch_async: cannot select device %d", m->device[i]);
            rc = R1_EHIP;
            break;
        }
        rc = r1_render_shard_device_batch(m->ctx[i], &q, n_frames, seed_stride, m->d_record[i], m->stream[i]);
    }
    if (rc != R1_OK)
        return drain(rc);
    ncclResult_t nr = m->rccl.GroupStart();
    if (nr != 0)
    {
        r1_set_error("ncclGroupStart failed: %s", m->rccl.GetErrorString(nr));
        return drain(R1_EHIP);
    }
    for (int i = 0; i < m->n && nr == 0; ++i) // [device][frame][record]
        nr = m->rccl.AllGather(m->d_record[i], m->d_gathered[i], record * (size_t)n_frames, R1_NCCL_UINT8, m->comm[i], m->stream[i]);
    const ncclResult_t ne = m->rccl.GroupEnd();
    if (nr != 0 || ne != 0)
    {
        r1_set_error("ncclAllGather of the batch's records failed: %s", m->rccl.GetErrorString(nr != 0 ? nr : ne));
        return drain(R1_EHIP);
    }
    hipError_t he = hipSetDevice(m->device[0]);
    rc = he == hipSuccess ? r1_assemble_device_records_batch(m->ctx[0], &p, n_frames, m->d_gathered[0], m->d_rgb, m->stream[0]) : R1_EHIP;
    if (rc == R1_OK && hipMemcpyAsync(host_frames, m->d_rgb, frame * (size_t)n_frames, hipMemcpyDeviceToHost, m->stream[0]) != hipSuccess)
    {
        r1_set_error("r1_multi_render_batch_async: copy to the host failed");
        rc = R1_EHIP;
    }
    return rc == R1_OK ? R1_OK : drain(rc);
}

extern "C" int r1_multi_render_async(r1_multi *m, const r1_params *params, void *host_frame)
{
    return r1_multi_render_batch_async(m, params, 1, 0u, host_frame);
}

extern "C" int r1_multi_sync(r1_multi *m)
{
    if (!m)
        return R1_EINVAL;
    for (int i = 0; i < m->n; ++i)
    {
        R1M_HIP(hipSetDevice(m->device[i]));
        R1M_HIP(hipStreamSynchronize(m->stream[i]));
    }
    return R1_OK;
}

extern "C" int r1_multi_info(r1_multi *m, int32_t *n_devices, int32_t *rccl_version, r1_launch_info *first_device)
{
    if (!m)
        return R1_EINVAL;
    if (n_devices)
        *n_devices = m->n;
    if (rccl_version)
        *rccl_version = m->rccl_version;
    if (first_device)
        return r1_last_launch_info(m->ctx[0], first_device);
    return R1_OK;
}
