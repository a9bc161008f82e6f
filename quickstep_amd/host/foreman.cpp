// foreman.cpp — QueryPlan, ForemanSingleNode and the Worker threads (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

// ---------------------------------------------------------------------------
// QueryPlan / Foreman / Worker
// ---------------------------------------------------------------------------
std::size_t QueryPlan::addRelationalOperator(RelationalOperator *op) {
  operators_.emplace_back(op);
  deps_.emplace_back();
  op->setOperatorIndex(operators_.size() - 1);
  return operators_.size() - 1;
}
void QueryPlan::addDirectDependency(std::size_t consumer, std::size_t producer, bool is_pipeline_breaker) {
  deps_.at(consumer).push_back(Edge{producer, is_pipeline_breaker});
}

ForemanSingleNode::ForemanSingleNode(QueryPlan *plan, QueryContext *query_context, StorageManager *storage_manager,
                                     std::size_t num_workers)
    : plan_(plan), query_context_(query_context), storage_manager_(storage_manager),
      num_workers_(num_workers ? num_workers : 1), outstanding_(plan->size(), 0), executing_(plan->size(), 0) {}

namespace {
// The Worker threads of the process (query_execution/Worker.hpp: created once at start-up, they outlive every query).
// Thread i keeps its HIP stream and with it everything that is cached per (thread, stream): the scratch arenas and staging
// buffers inside libqsx.so, this layer's cache of scratch allocations — a query does not pay for them again (creating
// four threads, streams and their first pinned / device buffers cost ~8 ms per query when the Foreman did it per run()).
// The threads are never joined: they sleep on their queues when the process exits.
class WorkerThreads {
 public:
  // The pool of one device (-1: no usable GPU, every work order fails with QSX_ERR_NO_DEVICE anyway).  One process per GPU is
  // the rule and then there is one pool; a process that drives several GPUs gets Worker threads — and streams — per device.
  static WorkerThreads &instance(int device) {
    static std::mutex pools_mutex;
    static std::map<int, WorkerThreads *> *pools = new std::map<int, WorkerThreads *>;
    std::lock_guard<std::mutex> lock(pools_mutex);
    WorkerThreads *&pool = (*pools)[device];
    if (pool == nullptr) pool = new WorkerThreads(device);
    return *pool;
  }
  // The device that is current in the calling thread: the one the caller's blocks, tables and states live on.
  static int callersDevice() {
    int device = -1;
    if (qsx_device_count() == 0 || qsx_current_device(&device) != QSX_OK) device = -1;
    return device;
  }
  // fn(i) on worker thread i for every i < n; returns when all have returned.
  void run(std::size_t n, const std::function<void(std::size_t)> &fn) {
    std::mutex done_mutex;
    std::condition_variable done_cv;
    std::size_t remaining = n;
    std::vector<Slot *> mine;   // (taken under the lock: another Foreman may be growing slots_ right now)
    {
      std::lock_guard<std::mutex> lock(mutex_);
      while (slots_.size() < n) {
        slots_.emplace_back(new Slot);
        Slot *slot = slots_.back().get();
        std::thread([slot, device = device_]() { threadMain(slot, device); }).detach();
      }
      for (std::size_t i = 0; i < n; ++i) mine.push_back(slots_[i].get());
    }
    for (std::size_t i = 0; i < n; ++i) {
      Slot *slot = mine[i];
      {
        std::lock_guard<std::mutex> lock(slot->mutex);
        slot->tasks.push_back([&, i]() {
          fn(i);
          std::lock_guard<std::mutex> done_lock(done_mutex);
          if (--remaining == 0) done_cv.notify_all();
        });
      }
      slot->cv.notify_one();
    }
    std::unique_lock<std::mutex> lock(done_mutex);
    done_cv.wait(lock, [&] { return remaining == 0; });
  }

 private:
  struct Slot {
    std::mutex mutex;
    std::condition_variable cv;
    std::deque<std::function<void()>> tasks;
  };
  explicit WorkerThreads(int device) : device_(device) {}
  static void threadMain(Slot *slot, int device) {
    qsx_stream_t stream = nullptr;
    // a new thread starts on device 0: move it to the pool's device before anything is created (stream, scratch, staging)
    if (device >= 0 && (qsx_set_current_device(device) != QSX_OK || qsx_stream_create(&stream) != QSX_OK)) {
      stream = nullptr;   // (the default stream then)
    }
    SetCurrentStream(stream);
    for (;;) {
      std::function<void()> task;
      {
        std::unique_lock<std::mutex> lock(slot->mutex);
        slot->cv.wait(lock, [&] { return !slot->tasks.empty(); });
        task = std::move(slot->tasks.front());
        slot->tasks.pop_front();
      }
      task();
    }
  }
  const int device_;
  std::mutex mutex_;
  std::vector<std::unique_ptr<Slot>> slots_;
};
}  // namespace

namespace {
thread_local bool tls_on_worker_thread = false;
}
void ForemanSingleNode::workerMain(std::size_t worker_id) {
  struct OnWorker {
    OnWorker() { tls_on_worker_thread = true; }
    ~OnWorker() { tls_on_worker_thread = false; }
  } on_worker;
  // Worker::run (query_execution/Worker.cpp:54-99): receive a work order, execute(), report completion.
  struct FreshStream {   // QSX_HOST_FRESH_STREAMS (debugging): a stream of this query only on the persistent thread
    qsx_stream_t previous = CurrentStream(), mine = nullptr;
    FreshStream() {
      if (std::getenv("QSX_HOST_FRESH_STREAMS") != nullptr && qsx_device_count() > 0 && qsx_stream_create(&mine) == QSX_OK) SetCurrentStream(mine);
    }
    ~FreshStream() {
      if (mine != nullptr) {
        qsx_stream_synchronize(mine);
        SetCurrentStream(previous);
        qsx_stream_destroy(mine);
      }
    }
  } fresh_stream;
  for (;;) {
    Item item;
    {
      std::unique_lock<std::mutex> lock(mutex_);
      // Work orders that want the device to themselves (WorkOrder::prefersExclusiveDevice) run only next to their own kind:
      // while one is queued nothing else starts, it starts when the others have drained, and the others resume when no
      // such work order is queued or running.  Nothing waits for a work order that is not already on a Worker.
      bool exclusive_queued = false;
      auto runnable = [&](const Item &it) {
        return it.exclusive ? shared_running_ == 0 : (exclusive_running_ == 0 && !exclusive_queued);
      };
      auto first_runnable = [&]() {
        exclusive_queued = false;
        for (const Item &it : ready_) exclusive_queued = exclusive_queued || it.exclusive;
        for (std::size_t i = 0; i < ready_.size(); ++i) {
          if (runnable(ready_[i])) return i;
        }
        return ready_.size();
      };
      std::size_t pick = 0;
      cv_work_.wait(lock, [&] {
        if (shutting_down_) return true;
        pick = first_runnable();
        return pick < ready_.size();
      });
      if (ready_.empty()) break;
      pick = first_runnable();
      if (pick == ready_.size()) break;   // (shutting down with work orders nobody may start: an error elsewhere)
      // The next work order: of the operator with the fewest work orders on Workers right now (first in the queue among
      // equals).  A probe whose build has just finished then starts next to an aggregation that queued seventy work orders
      // before it, instead of behind them: its host-side steps (counts read back, output blocks) overlap the other
      // operator's kernels.  (The reference's PolicyEnforcer picks per query; within one query it is FIFO.)
      for (std::size_t i = pick + 1; i < ready_.size() && executing_[ready_[pick].op] != 0; ++i) {
        if (runnable(ready_[i]) && executing_[ready_[i].op] < executing_[ready_[pick].op]) pick = i;
      }
      item = ready_[pick];
      ready_.erase(ready_.begin() + static_cast<std::ptrdiff_t>(pick));
      ++executing_[item.op];
      ++(item.exclusive ? exclusive_running_ : shared_running_);
    }
    std::unique_ptr<WorkOrder> wo(item.wo);
    const std::uint64_t start = NowMicros();
    std::string error;
    try {
      wo->execute();
    } catch (const std::exception &e) {
      error = e.what();
    }
    wo.reset();
    const std::uint64_t end = NowMicros();
    {
      std::lock_guard<std::mutex> lock(mutex_);
      --outstanding_[item.op];
      --executing_[item.op];
      --(item.exclusive ? exclusive_running_ : shared_running_);
      ++events_;
      profile_.push_back(WorkOrderTimeEntry{worker_id, item.op, start, end});
      if (!error.empty() && worker_error_.empty()) worker_error_ = error;
    }
    cv_work_.notify_all();   // (what may start depends on what runs)
    cv_done_.notify_all();
  }
}

void ForemanSingleNode::run() {
  // The process-wide Worker threads run one Foreman's workerMain at a time each: queries admitted concurrently from
  // different threads are served one after the other (the reference's Workers interleave the work orders of admitted
  // queries), and a Foreman started from INSIDE a work order would wait for the very thread it runs on.
  if (tls_on_worker_thread) {
    throw ExecutionError("ForemanSingleNode::run() called from a work order: nested query execution is not supported", QSX_ERR_UNSUPPORTED);
  }
  const std::size_t N = plan_->size();
  // Opt-in: measured on the headline plan (one 100 M-row probe next to 19 aggregation work orders) the probe alone takes
  // 0.95 ms instead of 4.2 ms of wall time next to the aggregation, but the step takes 5.2 ms either way — the device does
  // the same work in both orders and the drain before the probe costs what the undisturbed L2 gains.
  const char *exclusive_env = std::getenv("QSX_HOST_EXCLUSIVE_PROBES");
  const bool exclusive_probes = exclusive_env != nullptr && exclusive_env[0] == '1';
  WorkOrdersContainer container(N);
  std::vector<bool> done_generating(N, false), finished(N, false);
  std::vector<std::size_t> blocks_fed(N, 0);  // per producer: output blocks already fed downstream
  // the process-wide Worker threads serve this query until it shuts them out again (one more thread drives them and waits)
  const int device = WorkerThreads::callersDevice();
  std::thread workers([this, device]() {
    if (std::getenv("QSX_HOST_EPHEMERAL_WORKERS") != nullptr) {   // (debugging: threads and streams of this run() only)
      std::vector<std::thread> own;
      for (std::size_t w = 0; w < num_workers_; ++w) {
        own.emplace_back([this, w, device]() {
          qsx_stream_t stream = nullptr;
          if (device >= 0 && qsx_set_current_device(device) == QSX_OK) (void)qsx_stream_create(&stream);
          SetCurrentStream(stream);
          workerMain(w);
          if (stream != nullptr) qsx_stream_destroy(stream);
        });
      }
      for (auto &t : own) t.join();
      return;
    }
    WorkerThreads::instance(device).run(num_workers_, [this](std::size_t w) { workerMain(w); });
  });

  // The Foreman sleeps until something happened: a work order finished (workerMain) or an output block was registered with
  // one of the plan's destinations — the kDataPipelineMessage / kWorkOrderCompleteMessage the reference's Foreman blocks on
  // (query_execution/ForemanSingleNode.cpp:118-170), in place of a timed poll.
  std::vector<InsertDestination *> hooked;
  for (std::size_t op = 0; op < N; ++op) {
    const QueryContext::insert_destination_id dest_id = plan_->getOperator(op)->getInsertDestinationID();
    if (dest_id == QueryContext::kInvalidInsertDestinationId) continue;
    InsertDestination *dest = query_context_->getInsertDestination(dest_id);
    dest->setBlockReturnedCallback([this]() {
      {
        std::lock_guard<std::mutex> lock(mutex_);
        ++events_;
      }
      cv_done_.notify_all();
    });
    hooked.push_back(dest);
  }
  auto shutdown = [&]() {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      shutting_down_ = true;
    }
    cv_work_.notify_all();
    workers.join();
    for (InsertDestination *dest : hooked) dest->setBlockReturnedCallback(nullptr);
  };

  try {
    for (;;) {
      std::unique_lock<std::mutex> lock(mutex_);
      if (!worker_error_.empty()) throw std::runtime_error("work order failed: " + worker_error_);
      const std::uint64_t events_seen = events_;   // whatever happens from here on wakes the wait at the end of this pass
      bool progress = false;
      for (std::size_t op = 0; op < N; ++op) {
        if (finished[op]) continue;
        bool blocked = false, producers_finished = true;
        for (const QueryPlan::Edge &e : plan_->dependencies(op)) {
          if (!finished[e.producer]) {
            producers_finished = false;
            if (e.breaker) blocked = true;
          }
        }
        if (plan_->getOperator(op)->isCollective()) {
          // collectives in plan order, one operator at a time: every rank then issues the same sequence
          for (std::size_t earlier = 0; earlier < op; ++earlier) {
            if (!finished[earlier] && plan_->getOperator(earlier)->isCollective()) blocked = true;
          }
        }
        if (!blocked && !done_generating[op]) {
          // only the Foreman thread ever calls getAllWorkOrders (SURVEY §8b Threading)
          lock.unlock();
          const bool done = plan_->getOperator(op)->getAllWorkOrders(&container, query_context_, storage_manager_, 0, &bus_);
          lock.lock();
          while (WorkOrder *wo = container.getNormalWorkOrder(op)) {
            ready_.push_back(Item{wo, op, exclusive_probes && wo->prefersExclusiveDevice()});
            ++outstanding_[op];
            progress = true;
          }
          if (done) done_generating[op] = true;
        }
        // pipelining: feed newly produced output blocks to streaming consumers (kDataPipelineMessage)
        RelationalOperator *producer = plan_->getOperator(op);
        const QueryContext::insert_destination_id dest_id = producer->getInsertDestinationID();
        if (dest_id != QueryContext::kInvalidInsertDestinationId) {
          // (only the blocks that are new since the last pass are copied out of the destination)
          const std::vector<InsertDestination::TouchedBlock> fresh = query_context_->getInsertDestination(dest_id)->getTouchedBlocksSince(blocks_fed[op]);
          for (const InsertDestination::TouchedBlock &block : fresh) {
            ++blocks_fed[op];
            for (std::size_t consumer = 0; consumer < N; ++consumer) {
              for (const QueryPlan::Edge &e : plan_->dependencies(consumer)) {
                if (e.producer == op && !e.breaker) {
                  // kDataPipelineMessage carries the partition id of the block (InsertDestination.cpp:424-470)
                  plan_->getOperator(consumer)->feedInputBlock(block.id, producer->getOutputRelationID(), block.partition);
                  progress = true;
                }
              }
            }
          }
        }
        if (done_generating[op] && outstanding_[op] == 0 && producers_finished && container.getNumNormalWorkOrders(op) == 0) {
          // re-check that no block appeared between the scan above and now
          if (dest_id == QueryContext::kInvalidInsertDestinationId ||
              blocks_fed[op] == query_context_->getInsertDestination(dest_id)->numTouchedBlocks()) {
            finished[op] = true;
            progress = true;
            producer->updateCatalogOnCompletion();   // QueryManagerBase.cpp:184 (markOperatorFinished)
            for (std::size_t consumer = 0; consumer < N; ++consumer) {
              for (const QueryPlan::Edge &e : plan_->dependencies(consumer)) {
                if (e.producer == op && !e.breaker) {
                  plan_->getOperator(consumer)->doneFeedingInputBlocks(producer->getOutputRelationID());
                }
              }
            }
          }
        }
      }
      if (std::all_of(finished.begin(), finished.end(), [](bool f) { return f; })) break;
      if (progress) {
        lock.unlock();
        cv_work_.notify_all();
        continue;
      }
      cv_done_.wait(lock, [&] { return events_ != events_seen || !worker_error_.empty(); });
    }
  } catch (...) {
    shutdown();
    throw;
  }
  shutdown();
}

}  // namespace quickstep
