// ThreadSanitizer / stress check of host/worker_pool.hpp (run by tests/test_capi_host.py): thousands of small jobs with
// varying thread counts, every task executed exactly once, results visible to the caller after run() returns.
#include <cstdio>
#include <numeric>

#include "../../quick-adc_amd/host/worker_pool.hpp"

int main() {
    qadc::WorkerPool pool;
    std::vector<int> hits;
    long checksum = 0;
    for (int job = 0; job < 4000; ++job) {
        const int tasks = 1 + job % 37, threads = 1 + job % 9;
        hits.assign(tasks, 0);
        std::vector<long> out(tasks, 0);
        pool.run(tasks, threads, [&](int t) {
            hits[t] += 1;                                        // (each task index is handed out once: no two threads share t)
            long s = 0;
            for (int i = 0; i < 50 + t; ++i) s += i * (job + 1);
            out[t] = s;
        });
        for (int t = 0; t < tasks; ++t) {
            if (hits[t] != 1) { std::printf("job %d: task %d ran %d times\n", job, t, hits[t]); return 1; }
            checksum += out[t];
        }
    }
    std::printf("pool ok %ld\n", checksum);
    return 0;
}
