// supersonic_amd/sharded.h -- a C++ host's multi-GPU drivers for the sharded aggregates (BASELINE's metric at N > 1 and
// config #4), written directly against RCCL and the C ABI (include/ssgpu.h: "partial aggregates", "result images").
// One process (or thread) per GPU; rows are range-sharded; no input row crosses xGMI.
//
// The reference has no distributed code; the shape it documents for a sharded aggregation is
// aggregate-per-shard -> shuffle -> final aggregate (supersonic/cursor/core/aggregate.h:236-242).
//
// ShardedScalarAggregate (the headline query, Filter -> Compute -> ScalarAggregate):
//   the shard's rows up to the partial-aggregate state           ssgpu_plan_run_partial
//   -> ONE ncclAllGather of the state (a few hundred bytes)       RCCL over xGMI
//   -> ONE kernel folds the `world` states and emits the row      ssgpu_plan_fold_finalize
//
// ShardedGroupAggregate:
//   per-shard GroupAggregate (the caller's pipeline)            RunOnDevice()
//   -> its partial table packed into ONE device image           ssgpu_result_pack_image
//   -> ONE ncclAllGather of the images                          RCCL over xGMI
//   -> images laid out as contiguous columns + a validity flag  ssgpu_images_unpack
//   -> merge GroupAggregate (SUM of sums, MIN of mins, ...)     an ordinary cursor over ScanDeviceView
// or, with Exchange KEY_RANGE (the form that scales: every rank merges 1 / world of the key space and ends with the groups
// it owns, each link carries 1 / world of a table instead of all of it):
//   -> its partial table routed into `world` images by key      ssgpu_result_route_images
//   -> ONE all-to-all of the images (grouped ncclSend / ncclRecv: image d goes to rank d)
//   -> unpack + merge as above, over the rows this rank owns
//
// Everything between the two plan runs is stream-ordered on ONE stream (the context's): no device value is read on the
// host in between.  Both plans are bound ONCE and re-run on every Run() (DeviceCursor::Rewind): a step costs no binding,
// no allocation and no compilation.  DOUBLE sums cross shards as exact (SUM, SSGPU_SUM_RESIDUAL) pairs -- each group's
// double-double accumulator leaves its shard unrounded -- so the cross-shard total stays within 1 ULP of the exact sum
// (adding the shards' ROUNDED sums is hundreds to thousands of ULP off on ill-conditioned data, profiles/r03_double_sum_ulp.json).
// `supersonic_amd/distributed.py` is the same protocol over torch.distributed.  STRING columns need one dictionary for the
// whole job (distributed.py: job_strings); this driver builds none and refuses STRING keys / STRING results (ERROR_NOT_IMPLEMENTED).
#ifndef SUPERSONIC_AMD_SHARDED_H_
#define SUPERSONIC_AMD_SHARDED_H_

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <string>
#include <vector>

#include "supersonic.h"

namespace supersonic {

namespace internal {
// A cursor over a plan somebody else keeps (the sharded drivers re-run their plans every step and hand out the result of
// the LAST step): forwards to it, owns nothing.
class BorrowedCursor : public Cursor {
 public:
  explicit BorrowedCursor(DeviceCursor* c) : c_(c) {}
  const TupleSchema& schema() const override { return c_->schema(); }
  ResultView Next(rowcount_t max_row_count) override { return c_->Next(max_row_count); }
  void Interrupt() override { c_->Interrupt(); }
  void AppendDebugDescription(string* target) const override { c_->AppendDebugDescription(target); }
  CursorId GetCursorId() const override { return c_->GetCursorId(); }
  DeviceCursor* device_cursor() const { return c_; }
 private:
  DeviceCursor* c_;
};
inline FailureOrOwned<Cursor> FailCursor(int code, const std::string& message) { return FailureOrOwned<Cursor>(new Exception(code, message)); }
}  // namespace internal

// The headline query at N > 1: `ScalarAggregate(spec, child)` over the concatenation of every rank's `local_child`.
class ShardedScalarAggregate {
 public:
  // comm / world: the job's RCCL communicator and its size.  spec and local_child (this rank's shard, e.g.
  // Filter(..., Compute(..., ScanView(shard)))) are owned.
  ShardedScalarAggregate(ncclComm_t comm, int world, AggregationSpecification* spec, Operation* local_child)
      : comm_(comm), world_(world), op_(ScalarAggregate(spec, local_child)) {}
  ~ShardedScalarAggregate() { if (gathered_) (void)hipFree(gathered_); }

  // One step over this rank's rows, which are rows [global_row_offset, ...) of the job (FIRST / LAST follow the global row
  // order).  The returned cursor serves the ONE result row -- the same on every rank -- until the next Run().
  FailureOrOwned<Cursor> Run(int64_t global_row_offset) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    if (!cursor_) {
      FailureOrOwned<Cursor> c = op_->CreateCursor();
      if (c.is_failure()) return c;
      // a STRING result (MIN / MAX / FIRST / LAST of a STRING column) is a code of THIS plan's dictionary: the ranks' states must
      // not be folded (one dictionary for the whole job is not built here; cf. ShardedGroupAggregate)
      for (int i = 0; i < c->schema().attribute_count(); ++i)
        if (c->schema().attribute(i).type() == STRING)
          return internal::FailCursor(ERROR_NOT_IMPLEMENTED, "STRING aggregate results need one dictionary for the whole job: not available in this driver");
      cursor_.reset(c.release());
    }
    internal::DeviceCursor* dc = internal::AsDeviceCursor(cursor_.get());
    dc->Rewind();
    int rc = dc->RunPartialOnDevice(global_row_offset);
    if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
    // the partial-aggregate state: 8 arrays of n_slots 64-bit words, contiguous in segment order
    ssgpu_partial_segment segs[16];
    const int32_t n = ssgpu_plan_partial_segments(dc->plan_handle(), segs, 16);
    if (n <= 0) return internal::FailCursor(ERROR_NOT_IMPLEMENTED, "the plan exposes no partial-aggregate state");
    size_t words = 0;
    for (int32_t i = 0; i < n; ++i) {
      if (static_cast<const char*>(segs[i].device_ptr) != static_cast<const char*>(segs[0].device_ptr) + words * 8)
        return internal::FailCursor(ERROR_UNKNOWN_ERROR, "the partial-aggregate state is not contiguous");
      words += static_cast<size_t>(segs[i].count);
    }
    if (words * 8 * static_cast<size_t>(world_) > gathered_bytes_) {
      if (gathered_) (void)hipFree(gathered_);
      gathered_ = nullptr; gathered_bytes_ = 0;
      if (hipMalloc(&gathered_, words * 8 * static_cast<size_t>(world_)) != hipSuccess) return internal::FailCursor(ERROR_MEMORY_EXCEEDED, "cannot allocate the gathered states");
      gathered_bytes_ = words * 8 * static_cast<size_t>(world_);
    }
    hipStream_t stream = static_cast<hipStream_t>(ssgpu_ctx_stream(ctx));
    if (ncclAllGather(segs[0].device_ptr, gathered_, words * 8, ncclUint8, comm_, stream) != ncclSuccess)   // the ONE collective
      return internal::FailCursor(ERROR_UNKNOWN_ERROR, "ncclAllGather of the partial-aggregate states failed");
    rc = dc->FinalizePartial(gathered_, world_);
    if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
    return FailureOrOwned<Cursor>(new internal::BorrowedCursor(dc));
  }

 private:
  ncclComm_t comm_;
  int world_;
  std::unique_ptr<Operation> op_;
  std::unique_ptr<Cursor> cursor_;     // bound once, re-run every step
  void* gathered_ = nullptr;
  size_t gathered_bytes_ = 0;
};

class ShardedGroupAggregate {
 public:
  // DENSE (ssgpu.h "dense-slot GroupAggregate across ranks"): the ranks agree once on the value ranges of the group keys
  // (two small ncclAllReduce at set-up); from then on every rank's partial table is the same slot-indexed array and a step is
  // shard scan -> ONE all-to-all of slot slices -> element-wise fold + extraction on the SAME plan -- no merge plan, no packing,
  // DOUBLE sums cross as raw (hi, lo) accumulators.  Every rank ends with the groups it owns (as KEY_RANGE).  Plans it does not
  // fit (FIRST / LAST, floating keys, key ranges beyond 2^21 slots) fail Run() with the library's code, identically on every
  // rank: the caller constructs the job with KEY_RANGE instead.  capacity_rows is not used.
  enum Exchange { ALL_GATHER, KEY_RANGE, DENSE };
  // comm / world: the job's RCCL communicator and its size.  group_by: key attribute names.  spec and local_child
  // (this rank's shard: e.g. Filter(..., ScanView(shard))) are owned.  capacity_rows: rows an image holds -- at least
  // the largest partial table of any rank (a table that does not fit is reported by Run(), nothing is truncated silently).
  // With KEY_RANGE an image holds the rows ONE destination receives from ONE source (about 1.3 / world of a partial table,
  // the hash's spread included) and the result of Run() is this rank's share of the groups.
  ShardedGroupAggregate(ncclComm_t comm, int world, const std::vector<std::string>& group_by, AggregationSpecification* spec,
                        Operation* local_child, rowcount_t capacity_rows, Exchange exchange = ALL_GATHER)
      : comm_(comm), world_(world), group_by_(group_by), capacity_(capacity_rows), exchange_(exchange) {
    Init(spec, std::vector<Operation*>(1, local_child));
  }
  // Several shards resident on ONE device, no communicator: their images meet in this process as if they were further
  // ranks (a table larger than one block aggregated block by block -- and the way the N-rank arithmetic, e.g. the 1-ULP
  // property of cross-shard DOUBLE sums, is tested on a single GPU).  Same protocol, the collective left out.
  ShardedGroupAggregate(const std::vector<std::string>& group_by, AggregationSpecification* spec, const std::vector<Operation*>& local_children,
                        rowcount_t capacity_rows, Exchange exchange = ALL_GATHER)
      : comm_(nullptr), world_(1), group_by_(group_by), capacity_(capacity_rows), exchange_(exchange) {
    Init(spec, local_children);
  }
  ~ShardedGroupAggregate() { shard_cursors_.clear(); merge_cursor_.reset(); Free(); }

 private:
  void Init(AggregationSpecification* spec, const std::vector<Operation*>& local_children) {
    std::unique_ptr<AggregationSpecification> own_spec(spec);
    std::vector<std::unique_ptr<Operation>> own_children;
    for (Operation* c : local_children) own_children.emplace_back(c);
    if (own_children.empty()) { error_code_ = ERROR_INVALID_ARGUMENT_VALUE; error_ = "no local shard"; return; }
    // (a sharded job steps its plans without waiting for the host and keeps its shards' columns alive: its OWN plans opt in to
    // ssgpu.h's "lazy_feedback" when their cursors are made -- Run() -- and other plans of the shared context keep the default)
    if (exchange_ == DENSE) ssgpu_ctx_set_option(internal::Context::Get().ctx, "group_dense", 1);
    // the input's types decide which sums travel as (SUM, SUM_RESIDUAL) pairs: bind the child once to learn them
    TupleSchema child_schema;
    {
      FailureOrOwned<Cursor> probe = own_children[0]->CreateCursor();
      if (probe.is_failure()) { error_code_ = probe.exception().return_code(); error_ = probe.exception().message(); return; }
      child_schema = probe->schema();
    }
    // STRING values cross shards as the INT32 codes of a dictionary, and every plan builds its own from the strings it meets: a
    // job needs ONE dictionary for all its plans (distributed.py: job_strings, an all-gather of the ranks' strings at set-up).
    // This driver does not build one -- it refuses STRING keys and STRING results instead of merging codes of different
    // dictionaries (COUNT of a STRING column is a number and is fine).
    auto is_string = [&](const std::string& name) { const int pos = child_schema.LookupAttributePosition(name); return pos >= 0 && child_schema.attribute(pos).type() == STRING; };
    for (auto& k : group_by_) if (is_string(k)) { error_code_ = ERROR_NOT_IMPLEMENTED; error_ = "STRING group keys need one dictionary for the whole job: not available in this driver"; return; }
    for (auto& e : own_spec->elements)
      if (e.aggregation != COUNT && is_string(e.input)) { error_code_ = ERROR_NOT_IMPLEMENTED; error_ = "STRING aggregate results need one dictionary for the whole job: not available in this driver"; return; }
    // the shard's specification (+ residuals) and the merge functions of the aggregates (cf. distributed.py: _shard_spec, _merge_spec)
    std::unique_ptr<AggregationSpecification> shard(new AggregationSpecification), merged(new AggregationSpecification);
    for (auto& e : own_spec->elements) {
      if (e.distinct || e.aggregation == CONCAT) { error_code_ = ERROR_NOT_IMPLEMENTED; error_ = "aggregation cannot be merged across shards"; return; }
      shard->elements.push_back(e);
      const Aggregation m = e.aggregation == COUNT ? SUM : e.aggregation;
      merged->AddAggregation(m, e.output, e.output);
      if (e.aggregation == COUNT) counts_.push_back(e.output);
      const int pos = child_schema.LookupAttributePosition(e.input);
      // SUM / MIN / MAX / FIRST / LAST of a NOT NULL input are never NULL in a partial table (a group has a row): the merge reads
      // them as NOT NULL columns and keeps no contribution counts for them (cf. distributed.py: _never_null)
      const bool never_null = e.aggregation != COUNT && pos >= 0 && !child_schema.attribute(pos).is_nullable();
      if (never_null) never_null_.push_back(e.output);
      if (e.aggregation == SUM && pos >= 0 && (child_schema.attribute(pos).type() == DOUBLE || child_schema.attribute(pos).type() == FLOAT) &&
          (e.output_type == INT32 || e.output_type == UINT32 || e.output_type == INT64 || e.output_type == UINT64)) {
        // (the reference adds and truncates row after row, aggregation_operators.h:173-185: a shard's result is not a partial sum)
        error_code_ = ERROR_NOT_IMPLEMENTED; error_ = "SUM of a floating input into an integer output cannot be merged across shards"; return;
      }
      if (e.aggregation == SUM && pos >= 0 && child_schema.attribute(pos).type() == DOUBLE && (e.output_type < 0 || e.output_type == DOUBLE)) {
        shard->AddAggregation(static_cast<Aggregation>(SSGPU_SUM_RESIDUAL), e.input, e.output + kResidual);
        merged->AddAggregation(SUM, e.output + kResidual, e.output + kResidual);
        residuals_.push_back(e.output);
        if (never_null) never_null_.push_back(e.output + kResidual);
      }
    }
    merged_spec_ = std::move(merged);
    for (auto& child : own_children) {
      CompoundSingleSourceProjector* keys = new CompoundSingleSourceProjector;
      for (auto& k : group_by_) keys->add(ProjectNamedAttribute(k));
      // (DENSE: the job's own aggregation -- its accumulators travel raw, there is no merge plan to feed with residual columns)
      first_.emplace_back(GroupAggregate(keys, new AggregationSpecification(exchange_ == DENSE ? *own_spec : *shard), nullptr, child.release()));
    }
  }

 public:
  // One step.  On success the returned cursor serves the result of THIS step -- the full table (ALL_GATHER: the same on every
  // rank) or this rank's groups (KEY_RANGE) -- until the next Run().  On failure every rank fails: the verdict (a table
  // that did not fit, an evaluation error in any shard) is agreed on with one small all-reduce where the ranks saw
  // different images (KEY_RANGE), so no rank proceeds with a result the others rejected.
  FailureOrOwned<Cursor> Run() {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    if (!error_.empty()) return internal::FailCursor(error_code_, error_);
    while (shard_cursors_.size() < first_.size()) {
      FailureOrOwned<Cursor> shard = first_[shard_cursors_.size()]->CreateCursor();
      if (shard.is_failure()) return shard;
      if (internal::DeviceCursor* dc = internal::AsDeviceCursor(shard.get())) ssgpu_plan_set_option(dc->plan_handle(), "lazy_feedback", 1);
      shard_cursors_.emplace_back(shard.release());
    }
    const int n_local = static_cast<int>(shard_cursors_.size());
    const int n_images = comm_ ? world_ : n_local;                          // images that meet in this process's merge
    hipStream_t stream = static_cast<hipStream_t>(ssgpu_ctx_stream(ctx));   // NULL = the legacy default stream
    if (exchange_ == DENSE) return RunDense(ctx, stream);
    for (int attempt = 0; attempt < 6; ++attempt) {
      int rc = SSGPU_OK;
      // A rank whose shard run fails (memory quota, interrupt, an input that does not bind like the others') must NOT return here
      // while the other ranks enter the collective below -- they would wait in ncclAllGather / ncclSend forever.  It sends an
      // EMPTY image flagged with its return code instead (header word 5), and every rank fails the step with that code once
      // the trailer has been read.  (Without a communicator there is nobody to wait: fail at once.)
      int local_fail = SSGPU_OK; std::string local_fail_msg;
      for (int l = 0; l < n_local; ++l) {
        internal::DeviceCursor* sc = internal::AsDeviceCursor(shard_cursors_[static_cast<size_t>(l)].get());
        sc->Rewind();
        rc = sc->RunOnDevice();
        if (rc != SSGPU_OK) {
          if (!comm_) return internal::FailCursor(rc, ssgpu_last_error(ctx));
          local_fail = rc; local_fail_msg = ssgpu_last_error(ctx);
          break;
        }
      }
      internal::DeviceCursor* shard = internal::AsDeviceCursor(shard_cursors_[0].get());
      ssgpu_plan* plan = shard->plan_handle();
      const int n_attrs = ssgpu_plan_attr_count(plan);
      int64_t image_bytes = 0, unpacked_bytes = 0;
      rc = ssgpu_plan_image_layout(plan, static_cast<int64_t>(capacity_), n_images, &image_bytes, &unpacked_bytes, nullptr);
      if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
      if (image_bytes != image_bytes_ || unpacked_bytes != unpacked_bytes_) {
        merge_cursor_.reset(); merge_.reset();                       // (they scan the buffers that are about to move)
        Free();
        const size_t send_images = exchange_ == KEY_RANGE ? static_cast<size_t>(world_) : 1;     // key range: one image per destination
        if (hipMalloc(&image_, static_cast<size_t>(image_bytes) * send_images) != hipSuccess || hipMalloc(&images_, static_cast<size_t>(image_bytes) * n_images) != hipSuccess ||
            hipMalloc(&unpacked_, static_cast<size_t>(unpacked_bytes)) != hipSuccess || hipMalloc(&verdict_dev_, 4 * sizeof(int64_t)) != hipSuccess)
          return internal::FailCursor(ERROR_MEMORY_EXCEEDED, "cannot allocate the result images");
        image_bytes_ = image_bytes; unpacked_bytes_ = unpacked_bytes;
      }
      if (!comm_) {
        // shards of one device: every shard's image straight into its slot of the gathered buffer (with KEY_RANGE the one
        // destination is this process: routing with n_dest = 1 is the packing, in another row order)
        for (int l = 0; l < n_local && rc == SSGPU_OK; ++l) {
          internal::DeviceCursor* sc = internal::AsDeviceCursor(shard_cursors_[static_cast<size_t>(l)].get());
          char* slot = static_cast<char*>(images_) + static_cast<size_t>(l) * image_bytes;
          rc = exchange_ == KEY_RANGE ? ssgpu_result_route_images(sc->result_handle(), static_cast<int32_t>(group_by_.size()), 1, static_cast<int64_t>(capacity_), slot)
                                      : ssgpu_result_pack_image(sc->result_handle(), static_cast<int64_t>(capacity_), slot);
        }
        if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
      } else if (exchange_ == KEY_RANGE) {
        rc = local_fail == SSGPU_OK ? ssgpu_result_route_images(shard->result_handle(), static_cast<int32_t>(group_by_.size()), world_, static_cast<int64_t>(capacity_), image_) : local_fail;
        if (rc != SSGPU_OK) {
          if (local_fail == SSGPU_OK) { local_fail = rc; local_fail_msg = ssgpu_last_error(ctx); }
          if (!FlagFailedImages(world_, image_bytes, local_fail, stream)) return internal::FailCursor(ERROR_UNKNOWN_ERROR, "cannot flag the images of a failed shard run");
        }
        bool ok = ncclGroupStart() == ncclSuccess;                     // the ONE collective: image d -> rank d
        for (int r = 0; ok && r < world_; ++r)
          ok = ncclSend(static_cast<const char*>(image_) + static_cast<size_t>(r) * image_bytes, static_cast<size_t>(image_bytes), ncclUint8, r, comm_, stream) == ncclSuccess &&
               ncclRecv(static_cast<char*>(images_) + static_cast<size_t>(r) * image_bytes, static_cast<size_t>(image_bytes), ncclUint8, r, comm_, stream) == ncclSuccess;
        ok = (ncclGroupEnd() == ncclSuccess) && ok;
        if (!ok) return internal::FailCursor(ERROR_UNKNOWN_ERROR, "the all-to-all of the result images (ncclSend / ncclRecv) failed");
      } else {
        rc = local_fail == SSGPU_OK ? ssgpu_result_pack_image(shard->result_handle(), static_cast<int64_t>(capacity_), image_) : local_fail;
        if (rc != SSGPU_OK) {
          if (local_fail == SSGPU_OK) { local_fail = rc; local_fail_msg = ssgpu_last_error(ctx); }
          if (!FlagFailedImages(1, image_bytes, local_fail, stream)) return internal::FailCursor(ERROR_UNKNOWN_ERROR, "cannot flag the image of a failed shard run");
          rc = SSGPU_OK;
        }
        if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
        if (ncclAllGather(image_, images_, static_cast<size_t>(image_bytes), ncclUint8, comm_, stream) != ncclSuccess)   // the ONE collective
          return internal::FailCursor(ERROR_UNKNOWN_ERROR, "ncclAllGather failed");
      }
      if (!merge_cursor_) {
        gathered_.schema = TupleSchema();
        for (int i = 0; i < n_attrs; ++i) {
          ssgpu_attr a; ssgpu_plan_attr(plan, i, &a);
          gathered_.schema.add_attribute(Attribute(a.name, static_cast<DataType>(a.dtype), Has(never_null_, a.name) ? NOT_NULLABLE : static_cast<Nullability>(a.nullable)));
        }
        gathered_.schema.add_attribute(Attribute("__valid", BOOL, NOT_NULLABLE));
        gathered_.columns.assign(static_cast<size_t>(n_attrs) + 1, ssgpu_column());
      }
      rc = ssgpu_images_unpack(plan, images_, n_images, static_cast<int64_t>(capacity_), unpacked_, gathered_.columns.data());
      if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
      gathered_.row_count = static_cast<rowcount_t>(n_images) * capacity_;
      if (!merge_cursor_) {
        // the merge, bound once: GroupAggregate of the merge functions over the real rows; COUNT columns back to NOT NULL,
        // every DOUBLE sum = its merged sum + its merged residual (each accumulated in double-double), residuals projected away
        CompoundSingleSourceProjector* keys = new CompoundSingleSourceProjector;
        for (auto& k : group_by_) keys->add(ProjectNamedAttribute(k));
        std::unique_ptr<AggregationSpecification> ms(new AggregationSpecification(*merged_spec_));
        Operation* merge = GroupAggregate(keys, ms.release(), nullptr, Filter(NamedAttribute("__valid"), ProjectAllAttributes(), ScanDeviceView(gathered_)));
        if (!counts_.empty() || !residuals_.empty()) {
          CompoundExpression* e = new CompoundExpression;
          for (int i = 0; i < n_attrs; ++i) {
            const Attribute& a = gathered_.schema.attribute(i);
            if (a.name().size() > strlen(kResidual) && a.name().compare(a.name().size() - strlen(kResidual), strlen(kResidual), kResidual) == 0) continue;
            if (Has(counts_, a.name())) e->AddAs(a.name(), IfNull(NamedAttribute(a.name()), a.type() == UINT64 ? ConstUint64(0) : ConstUint32(0)));
            else if (Has(residuals_, a.name())) e->AddAs(a.name(), Plus(NamedAttribute(a.name()), NamedAttribute(a.name() + kResidual)));
            else e->Add(NamedAttribute(a.name()));
          }
          merge = Compute(e, merge);
        }
        merge_.reset(merge);
        FailureOrOwned<Cursor> mc = merge_->CreateCursor();
        if (mc.is_failure()) return mc;
        if (internal::DeviceCursor* dc = internal::AsDeviceCursor(mc.get())) ssgpu_plan_set_option(dc->plan_handle(), "lazy_feedback", 1);
        merge_cursor_.reset(mc.release());
      }
      internal::DeviceCursor* merge = internal::AsDeviceCursor(merge_cursor_.get());
      merge->Rewind();
      rc = merge->RunOnDevice();
      if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
      // the final check, once per step and after everything was enqueued: did every table fit, did any shard fail?
      // trailer = {largest table, sum of row counts, any image flagged (full, or its shard has to repeat the step), OR of error words}
      int64_t trailer[4] = {0, 0, 0, 0};
      const char* trailer_dev = static_cast<const char*>(unpacked_) + unpacked_bytes_ - 32;
      if (comm_ && exchange_ == KEY_RANGE && world_ > 1) {
        // every rank unpacked DIFFERENT images: agree on the verdict (MAX over ranks of every word) before anybody decides
        if (hipMemcpyAsync(verdict_dev_, trailer_dev, 32, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
            ncclAllReduce(verdict_dev_, verdict_dev_, 4, ncclInt64, ncclMax, comm_, stream) != ncclSuccess)
          return internal::FailCursor(ERROR_UNKNOWN_ERROR, "cannot agree on the step's verdict (ncclAllReduce)");
        trailer_dev = static_cast<const char*>(verdict_dev_);
      }
      if (hipMemcpyAsync(trailer, trailer_dev, 32, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return internal::FailCursor(ERROR_UNKNOWN_ERROR, "cannot read the images' trailer");
      largest_table_ = static_cast<rowcount_t>(trailer[0]);
      // (every rank reads the same trailer -- after the all-reduce above where they unpacked different images -- so every rank
      //  takes the same branch below: nobody proceeds with a result the others rejected, nobody repeats alone)
      if (trailer[3] >> 8) {
        const int code = static_cast<int>(trailer[3] >> 8);
        return internal::FailCursor(code, local_fail != SSGPU_OK ? local_fail_msg : "the GroupAggregate of another rank's shard failed");
      }
      if (trailer[3] & 0xFF) return internal::FailCursor(ERROR_EVALUATION_ERROR, "Evaluation error in a shard's GroupAggregate");
      if (trailer[2]) {
        if (attempt + 1 >= 6) break;
        // a table that outgrew the images: larger images (as distributed.py's check() does) and the step again; flagged without
        // one: a shard's lazily read run feedback asked for the step to be repeated (ssgpu.h, ssgpu_result_route_images)
        if (largest_table_ > capacity_) capacity_ = (largest_table_ * 5 / 4 + 1023) / 1024 * 1024;
        continue;
      }
      return FailureOrOwned<Cursor>(new internal::BorrowedCursor(merge));
    }
    return internal::FailCursor(ERROR_UNKNOWN_ERROR, "the step kept asking to be repeated");
  }
  rowcount_t largest_table() const { return largest_table_; }   // rows of the largest partial table (KEY_RANGE: image) seen by the last Run()
  int64_t dense_slots() const { return dense_layout_.slots; }     // DENSE: slots of the job-wide table (0 before the first Run())

 private:
  // ---- the DENSE exchange ---------------------------------------------------------------------------------------------------------
  // the ranks' key ranges united: lo = min, hi = max over the ranks (order-preserving unsigned domain, ssgpu.h)
  bool AgreeRanges(int32_t n_keys, uint64_t* lo, uint64_t* hi, hipStream_t stream) {
    if (!comm_ || world_ == 1) return true;
    const size_t bytes = static_cast<size_t>(n_keys) * 8;
    if (!ranges_dev_ && hipMalloc(&ranges_dev_, 2 * 8 * 8) != hipSuccess) return false;
    char* d = static_cast<char*>(ranges_dev_);
    return hipMemcpyAsync(d, lo, bytes, hipMemcpyHostToDevice, stream) == hipSuccess && hipMemcpyAsync(d + 64, hi, bytes, hipMemcpyHostToDevice, stream) == hipSuccess &&
           ncclAllReduce(d, d, static_cast<size_t>(n_keys), ncclUint64, ncclMin, comm_, stream) == ncclSuccess &&
           ncclAllReduce(d + 64, d + 64, static_cast<size_t>(n_keys), ncclUint64, ncclMax, comm_, stream) == ncclSuccess &&
           hipMemcpyAsync(lo, d, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess && hipMemcpyAsync(hi, d + 64, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess &&
           hipStreamSynchronize(stream) == hipSuccess;
  }
  FailureOrOwned<Cursor> SetUpDense(internal::DeviceCursor* shard, hipStream_t stream) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    int32_t n_keys = 0; uint64_t lo[8], hi[8];
    int rc = shard->DenseKeyRanges(&n_keys, lo, hi);
    // (refusals are decided by the plan's shape -- the same on every rank -- before any collective: no rank is left waiting)
    if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
    if (dense_ready_) for (int32_t k = 0; k < n_keys; ++k) { lo[k] = std::min(lo[k], dense_lo_[k]); hi[k] = std::max(hi[k], dense_hi_[k]); }   // never narrower
    if (!AgreeRanges(n_keys, lo, hi, stream)) return internal::FailCursor(ERROR_UNKNOWN_ERROR, "cannot agree on the key ranges (ncclAllReduce)");
    rc = shard->SetDense(n_keys, lo, hi, comm_ ? world_ : 1, &dense_layout_);
    if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
    for (int32_t k = 0; k < n_keys; ++k) { dense_lo_[k] = lo[k]; dense_hi_[k] = hi[k]; }
    const size_t bytes = static_cast<size_t>(dense_layout_.chunk_bytes) * static_cast<size_t>(comm_ ? world_ : 1);
    if (bytes != dense_bytes_) {
      if (table_) (void)hipFree(table_);
      if (chunks_) (void)hipFree(chunks_);
      table_ = chunks_ = nullptr;
      if (hipMalloc(&table_, bytes) != hipSuccess || hipMalloc(&chunks_, bytes) != hipSuccess) return internal::FailCursor(ERROR_MEMORY_EXCEEDED, "cannot allocate the dense table buffers");
      dense_bytes_ = bytes;
    }
    dense_ready_ = true;
    return FailureOrOwned<Cursor>(static_cast<Cursor*>(nullptr));
  }
  FailureOrOwned<Cursor> RunDense(ssgpu_ctx* ctx, hipStream_t stream) {
    if (shard_cursors_.size() != 1) return internal::FailCursor(ERROR_NOT_IMPLEMENTED, "the DENSE exchange takes one shard per process");
    internal::DeviceCursor* shard = internal::AsDeviceCursor(shard_cursors_[0].get());
    const int n = comm_ ? world_ : 1;
    for (int attempt = 0; attempt < 8; ++attempt) {
      if (!dense_ready_) { FailureOrOwned<Cursor> s = SetUpDense(shard, stream); if (s.is_failure()) return s; }
      shard->Rewind();
      int rc = shard->RunDense(table_);
      if (rc != SSGPU_OK) {
        // this rank's run failed: it still joins the all-to-all -- with flagged chunks -- and every rank fails the step below
        const std::string msg = ssgpu_last_error(ctx);
        if (ssgpu_plan_dense_fail(shard->plan_handle(), table_, rc) != SSGPU_OK) return internal::FailCursor(rc, msg);
        local_fail_msg_ = msg;
      }
      if (comm_) {
        bool ok = ncclGroupStart() == ncclSuccess;                     // the ONE collective: chunk r -> rank r
        for (int r = 0; ok && r < world_; ++r)
          ok = ncclSend(static_cast<const char*>(table_) + static_cast<size_t>(r) * dense_layout_.chunk_bytes, static_cast<size_t>(dense_layout_.chunk_bytes), ncclUint8, r, comm_, stream) == ncclSuccess &&
               ncclRecv(static_cast<char*>(chunks_) + static_cast<size_t>(r) * dense_layout_.chunk_bytes, static_cast<size_t>(dense_layout_.chunk_bytes), ncclUint8, r, comm_, stream) == ncclSuccess;
        ok = (ncclGroupEnd() == ncclSuccess) && ok;
        if (!ok) return internal::FailCursor(ERROR_UNKNOWN_ERROR, "the all-to-all of the dense table chunks (ncclSend / ncclRecv) failed");
      }
      rc = shard->FoldDense(comm_ ? chunks_ : table_, n);
      if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
      // the verdict every rank reads from the headers it received: the same words everywhere, so the same branch everywhere
      uint32_t flags = 0, error = 0;
      rc = ssgpu_plan_dense_flags(shard->plan_handle(), &flags, &error);
      if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx));
      if (flags & 8u) return internal::FailCursor(static_cast<int>(flags >> 8), local_fail_msg_.empty() ? "the GroupAggregate of another rank's shard failed" : local_fail_msg_);
      if (error) return internal::FailCursor(ERROR_EVALUATION_ERROR, "Evaluation error in a shard's GroupAggregate");
      if (flags & 4u) { dense_ready_ = false; continue; }       // a key outside the ranges on some rank: agree on wider ones
      if (flags & 2u) { rc = ssgpu_plan_dense_grow(shard->plan_handle()); if (rc != SSGPU_OK) return internal::FailCursor(rc, ssgpu_last_error(ctx)); continue; }
      if (flags & 1u) return internal::FailCursor(ERROR_MEMORY_EXCEEDED, "a dense partition outgrew its table");
      return FailureOrOwned<Cursor>(new internal::BorrowedCursor(shard));
    }
    return internal::FailCursor(ERROR_UNKNOWN_ERROR, "the dense step kept asking to be repeated");
  }
 public:

 private:
  static constexpr const char* kResidual = "$res";   // suffix of the hidden column that carries a DOUBLE sum's residual across shards
  static bool Has(const std::vector<std::string>& v, const std::string& x) { for (auto& e : v) if (e == x) return true; return false; }
  // header (64 bytes: rows, capacity, flagged, rows wanted, evaluation errors, FAILURE CODE, 0, 0) of n empty images of a failed run
  bool FlagFailedImages(int n, int64_t image_bytes, int code, hipStream_t stream) {
    fail_header_[0] = 0; fail_header_[1] = static_cast<int64_t>(capacity_); fail_header_[2] = 0; fail_header_[3] = 0; fail_header_[4] = 0;
    fail_header_[5] = code; fail_header_[6] = 0; fail_header_[7] = 0;
    for (int d = 0; d < n; ++d)
      if (hipMemcpyAsync(static_cast<char*>(image_) + static_cast<size_t>(d) * image_bytes, fail_header_, sizeof(fail_header_), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
    return hipStreamSynchronize(stream) == hipSuccess;     // (fail_header_ is pageable host memory: the copies are done before it changes)
  }
  void Free() {
    if (image_) (void)hipFree(image_);
    if (images_) (void)hipFree(images_);
    if (unpacked_) (void)hipFree(unpacked_);
    if (verdict_dev_) (void)hipFree(verdict_dev_);
    if (table_) (void)hipFree(table_);
    if (chunks_) (void)hipFree(chunks_);
    if (ranges_dev_) (void)hipFree(ranges_dev_);
    table_ = chunks_ = ranges_dev_ = nullptr; dense_bytes_ = 0;
    image_ = images_ = unpacked_ = verdict_dev_ = nullptr; image_bytes_ = unpacked_bytes_ = 0;
  }
  ncclComm_t comm_;
  int world_;
  std::vector<std::string> group_by_, counts_, residuals_, never_null_;
  rowcount_t capacity_, largest_table_ = 0;
  Exchange exchange_ = ALL_GATHER;
  std::string error_;
  int error_code_ = OK;
  std::vector<std::unique_ptr<Operation>> first_;          // one per local shard (one, unless the shards of one device are simulated ranks)
  std::unique_ptr<Operation> merge_;
  std::vector<std::unique_ptr<Cursor>> shard_cursors_;     // bound once, re-run every step (declared after the operations: destroyed first)
  std::unique_ptr<Cursor> merge_cursor_;
  std::unique_ptr<AggregationSpecification> merged_spec_;
  DeviceView gathered_;
  void* image_ = nullptr; void* images_ = nullptr; void* unpacked_ = nullptr; void* verdict_dev_ = nullptr;
  int64_t image_bytes_ = 0, unpacked_bytes_ = 0;
  int64_t fail_header_[8] = {0};
  // DENSE
  bool dense_ready_ = false;
  ssgpu_dense_layout dense_layout_ = {};
  uint64_t dense_lo_[8] = {0}, dense_hi_[8] = {0};
  void* table_ = nullptr; void* chunks_ = nullptr; void* ranges_dev_ = nullptr;
  size_t dense_bytes_ = 0;
  std::string local_fail_msg_;
};

}  // namespace supersonic
#endif  // SUPERSONIC_AMD_SHARDED_H_
