// supersonic_amd/sharded.h -- a C++ host's multi-GPU driver for the sharded GroupAggregate (BASELINE config #4), written
// directly against RCCL and the C ABI's result images (include/ssgpu.h, "result images").  One process (or thread) per
// GPU; rows are range-sharded; no input row crosses xGMI.
//
// The reference has no distributed code; the shape it documents for a sharded aggregation is
// aggregate-per-shard -> shuffle -> final aggregate (supersonic/cursor/core/aggregate.h:236-242).  Here:
//
//   per-shard GroupAggregate (the caller's pipeline)            RunOnDevice()
//   -> its partial table packed into ONE device image           ssgpu_result_pack_image
//   -> ONE ncclAllGather of the images                          RCCL over xGMI
//   -> images laid out as contiguous columns + a validity flag  ssgpu_images_unpack
//   -> merge GroupAggregate (SUM of sums, MIN of mins, ...)     an ordinary cursor over ScanDeviceView
//
// or, with Exchange KEY_RANGE (the form that scales: every rank merges 1 / world of the key space and ends with the groups
// it owns, each link carries 1 / world of a table instead of all of it):
//
//   -> its partial table routed into `world` images by key      ssgpu_result_route_images
//   -> ONE all-to-all of the images (grouped ncclSend / ncclRecv: image d goes to rank d)
//   -> unpack + merge as above, over the rows this rank owns
//
// Everything between the two plan runs is stream-ordered on ONE stream (the context's): no device value is read on the
// host in between.  `supersonic_amd/distributed.py: DeviceShardedGroupAggregate` is the same protocol over
// torch.distributed; this header is for hosts that link RCCL themselves.  Like the reference's cursors, the mirror's
// Cursor is single-shot, so Run() binds its two cursors anew on every call (cheap next to a shard's aggregation; the image
// buffers are kept); a host that steps the same job thousands of times per second keeps the two ssgpu_plan handles and calls
// ssgpu_plan_run on them, as distributed.py does.  STRING columns need one dictionary for the whole job
// (distributed.py: job_strings) and are not handled here.
#ifndef SUPERSONIC_AMD_SHARDED_H_
#define SUPERSONIC_AMD_SHARDED_H_

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <string>
#include <vector>

#include "supersonic.h"

namespace supersonic {

class ShardedGroupAggregate {
 public:
  enum Exchange { ALL_GATHER, KEY_RANGE };
  // comm / world: the job's RCCL communicator and its size.  group_by: key attribute names.  spec and local_child
  // (this rank's shard: e.g. Filter(..., ScanView(shard))) are owned.  capacity_rows: rows an image holds -- at least
  // the largest partial table of any rank (a table that does not fit is reported by Run(), nothing is truncated silently).
  // With KEY_RANGE an image holds the rows ONE destination receives from ONE source (about 1.3 / world of a partial table,
  // the hash's spread included) and the result of Run() is this rank's share of the groups.
  ShardedGroupAggregate(ncclComm_t comm, int world, const std::vector<std::string>& group_by, AggregationSpecification* spec,
                        Operation* local_child, rowcount_t capacity_rows, Exchange exchange = ALL_GATHER)
      : comm_(comm), world_(world), group_by_(group_by), capacity_(capacity_rows), exchange_(exchange) {
    // the merge functions of the aggregates (cf. distributed.py: _merge_spec)
    std::unique_ptr<AggregationSpecification> merged(new AggregationSpecification);
    for (auto& e : spec->elements) {
      if (e.distinct || e.aggregation == CONCAT) { error_ = "aggregation cannot be merged across shards"; break; }
      const Aggregation m = e.aggregation == COUNT ? SUM : e.aggregation;
      merged->AddAggregation(m, e.output, e.output);
      if (e.aggregation == COUNT) counts_.push_back(e.output);
    }
    merged_spec_ = std::move(merged);
    CompoundSingleSourceProjector* keys = new CompoundSingleSourceProjector;
    for (auto& k : group_by) keys->add(ProjectNamedAttribute(k));
    first_.reset(GroupAggregate(keys, spec, nullptr, local_child));
  }
  ~ShardedGroupAggregate() { Free(); }

  // One step.  On success the returned cursor serves the full result (the same on every rank).
  FailureOrOwned<Cursor> Run() {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    if (!error_.empty()) return Fail(ERROR_NOT_IMPLEMENTED, error_);
    FailureOrOwned<Cursor> shard = first_->CreateCursor();
    if (shard.is_failure()) return Fail(shard.exception().return_code(), shard.exception().message());
    int rc = shard->RunOnDevice();
    if (rc != SSGPU_OK) return Fail(rc, ssgpu_last_error(ctx));
    ssgpu_plan* plan = shard->plan_handle();
    const int n_attrs = ssgpu_plan_attr_count(plan);
    int64_t image_bytes = 0, unpacked_bytes = 0;
    rc = ssgpu_plan_image_layout(plan, capacity_, world_, &image_bytes, &unpacked_bytes, nullptr);
    if (rc != SSGPU_OK) return Fail(rc, ssgpu_last_error(ctx));
    if (image_bytes != image_bytes_ || unpacked_bytes != unpacked_bytes_) {
      Free();
      const size_t send_images = exchange_ == KEY_RANGE ? static_cast<size_t>(world_) : 1;     // key range: one image per destination
      if (hipMalloc(&image_, static_cast<size_t>(image_bytes) * send_images) != hipSuccess || hipMalloc(&images_, static_cast<size_t>(image_bytes) * world_) != hipSuccess ||
          hipMalloc(&unpacked_, static_cast<size_t>(unpacked_bytes)) != hipSuccess)
        return Fail(ERROR_MEMORY_EXCEEDED, "cannot allocate the result images");
      image_bytes_ = image_bytes; unpacked_bytes_ = unpacked_bytes;
    }
    hipStream_t stream = static_cast<hipStream_t>(ssgpu_ctx_stream(ctx));   // NULL = the legacy default stream
    if (exchange_ == KEY_RANGE) {
      rc = ssgpu_result_route_images(shard->result_handle(), static_cast<int32_t>(group_by_.size()), world_, capacity_, image_);
      if (rc != SSGPU_OK) return Fail(rc, ssgpu_last_error(ctx));
      bool ok = ncclGroupStart() == ncclSuccess;                     // the ONE collective: image d -> rank d
      for (int r = 0; ok && r < world_; ++r)
        ok = ncclSend(static_cast<const char*>(image_) + static_cast<size_t>(r) * image_bytes, static_cast<size_t>(image_bytes), ncclUint8, r, comm_, stream) == ncclSuccess &&
             ncclRecv(static_cast<char*>(images_) + static_cast<size_t>(r) * image_bytes, static_cast<size_t>(image_bytes), ncclUint8, r, comm_, stream) == ncclSuccess;
      ok = (ncclGroupEnd() == ncclSuccess) && ok;
      if (!ok) return Fail(ERROR_UNKNOWN_ERROR, "the all-to-all of the result images (ncclSend / ncclRecv) failed");
    } else {
      rc = ssgpu_result_pack_image(shard->result_handle(), capacity_, image_);
      if (rc != SSGPU_OK) return Fail(rc, ssgpu_last_error(ctx));
      if (ncclAllGather(image_, images_, static_cast<size_t>(image_bytes), ncclUint8, comm_, stream) != ncclSuccess)   // the ONE collective
        return Fail(ERROR_UNKNOWN_ERROR, "ncclAllGather failed");
    }
    gathered_.schema = TupleSchema();
    for (int i = 0; i < n_attrs; ++i) {
      ssgpu_attr a; ssgpu_plan_attr(plan, i, &a);
      gathered_.schema.add_attribute(Attribute(a.name, static_cast<DataType>(a.dtype), static_cast<Nullability>(a.nullable)));
    }
    gathered_.schema.add_attribute(Attribute("__valid", BOOL, NOT_NULLABLE));
    gathered_.columns.assign(static_cast<size_t>(n_attrs) + 1, ssgpu_column());
    rc = ssgpu_images_unpack(plan, images_, world_, capacity_, unpacked_, gathered_.columns.data());
    if (rc != SSGPU_OK) return Fail(rc, ssgpu_last_error(ctx));
    gathered_.row_count = static_cast<rowcount_t>(world_) * capacity_;
    // the merge: GroupAggregate of the merge functions over the real rows (+ COUNT columns back to NOT NULL)
    CompoundSingleSourceProjector* keys = new CompoundSingleSourceProjector;
    for (auto& k : group_by_) keys->add(ProjectNamedAttribute(k));
    std::unique_ptr<AggregationSpecification> ms(new AggregationSpecification(*merged_spec_));
    Operation* merge = GroupAggregate(keys, ms.release(), nullptr,
                                      Filter(NamedAttribute("__valid"), ProjectAllAttributes(), ScanDeviceView(gathered_)));
    if (!counts_.empty()) {
      CompoundExpression* e = new CompoundExpression;
      for (int i = 0; i < n_attrs; ++i) {
        const Attribute& a = gathered_.schema.attribute(i);
        bool is_count = false;
        for (auto& cname : counts_) is_count = is_count || cname == a.name();
        if (is_count) e->AddAs(a.name(), IfNull(NamedAttribute(a.name()), a.type() == UINT64 ? ConstUint64(0) : ConstUint32(0)));
        else e->Add(NamedAttribute(a.name()));
      }
      merge = Compute(e, merge);
    }
    merge_.reset(merge);
    FailureOrOwned<Cursor> result = merge_->CreateCursor();
    if (result.is_failure()) return result;
    rc = result->RunOnDevice();
    if (rc != SSGPU_OK) return Fail(rc, ssgpu_last_error(ctx));
    // the final check, once per step and after everything was enqueued: did every table fit, did any shard fail?
    int64_t trailer[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(trailer, static_cast<const char*>(unpacked_) + unpacked_bytes_ - 32, 32, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return Fail(ERROR_UNKNOWN_ERROR, "cannot read the images' trailer");
    largest_table_ = trailer[0];
    if (trailer[3]) return Fail(ERROR_EVALUATION_ERROR, "Evaluation error in a shard's GroupAggregate");
    if (trailer[2]) return Fail(ERROR_MEMORY_EXCEEDED, "a shard's partial table has more rows than capacity_rows (largest_table() tells how many)");
    return result;
  }
  rowcount_t largest_table() const { return largest_table_; }   // rows of the largest partial table seen by the last Run()

 private:
  static FailureOrOwned<Cursor> Fail(int code, const std::string& message) { return FailureOrOwned<Cursor>(new Exception(code, message)); }
  void Free() {
    if (image_) (void)hipFree(image_);
    if (images_) (void)hipFree(images_);
    if (unpacked_) (void)hipFree(unpacked_);
    image_ = images_ = unpacked_ = nullptr; image_bytes_ = unpacked_bytes_ = 0;
  }
  ncclComm_t comm_;
  int world_;
  std::vector<std::string> group_by_, counts_;
  rowcount_t capacity_, largest_table_ = 0;
  Exchange exchange_ = ALL_GATHER;
  std::string error_;
  std::unique_ptr<Operation> first_, merge_;
  std::unique_ptr<AggregationSpecification> merged_spec_;
  DeviceView gathered_;
  void* image_ = nullptr; void* images_ = nullptr; void* unpacked_ = nullptr;
  int64_t image_bytes_ = 0, unpacked_bytes_ = 0;
};

}  // namespace supersonic
#endif  // SUPERSONIC_AMD_SHARDED_H_
