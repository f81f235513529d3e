// Host-side top-R heap of the MI355X Quick-ADC engine (C++14, header only).
//
// Same observable behaviour as the reference's kv_binheap<Key,Value> (binheap.hpp:18-142): the
// candidate stream returned by the device is replayed through this heap, and the resulting
// arrays (keys()[i], values()[i], size()) are bit-identical to what the reference's sequential
// scan leaves in its heap.  Written independently of the oracle's C restatement.
#pragma once
#include <algorithm>
#include <memory>
#include <utility>
#include <vector>

namespace qadc {

template <typename Key, typename Value>
class kv_heap {
public:
    typedef Key key_type;
    typedef Value value_type;

    kv_heap() : cap_(0), size_(0) {}
    explicit kv_heap(int capacity) : keys_(capacity), vals_(capacity), cap_(capacity), size_(0) {}

    void reset_capacity(int capacity) {
        keys_.assign(capacity, Key());
        vals_.assign(capacity, Value());
        cap_ = capacity;
        size_ = 0;
    }
    void reset() { size_ = 0; }

    int capacity() const { return cap_; }
    int size() const { return size_; }
    Value max() const { return vals_[0]; }           // caller guarantees size() >= 1 (binheap.hpp:63-65)
    // adopt an existing heap array (size <= capacity entries already in heap order)
    void assign(const Key* keys, const Value* values, int size) {
        size_ = size;
        for (int i = 0; i < size; ++i) { keys_[i] = keys[i]; vals_[i] = values[i]; }
    }
    const Key* keys() const { return keys_.data(); }
    const Value* values() const { return vals_.data(); }

    // binheap.hpp:75-116.  While the heap has room every push is accepted (appended and bubbled up
    // past strictly smaller parents).  Once full, a push is accepted only if strictly smaller than
    // the root; it then sinks, preferring the left child on ties, stopping at a child <= itself.
    void push(Key key, Value value) {
        if (size_ != cap_) {
            int i = size_++;
            while (i != 0) {
                const int parent = (i - 1) / 2;
                if (!(value > vals_[parent])) break;
                vals_[i] = vals_[parent];
                keys_[i] = keys_[parent];
                i = parent;
            }
            vals_[i] = value;
            keys_[i] = key;
            return;
        }
        if (!(value < vals_[0])) return;
        int i = 0;
        for (;;) {
            const int l = 2 * i + 1;
            if (l >= size_) break;
            int c = l;
            if (l + 1 < size_ && vals_[l + 1] > vals_[l]) c = l + 1;
            if (vals_[c] <= value) break;
            vals_[i] = vals_[c];
            keys_[i] = keys_[c];
            i = c;
        }
        vals_[i] = value;
        keys_[i] = key;
    }

    // binheap.hpp:118-137: ascending by value through std::sort on a permutation (tie order is
    // whatever std::sort yields, as in the reference).
    void sort(Key* out_keys, Value* out_values) const {
        std::vector<int> perm(size_);
        for (int i = 0; i < size_; ++i) perm[i] = i;
        std::sort(perm.begin(), perm.end(), [this](int a, int b) { return vals_[a] < vals_[b]; });
        for (int i = 0; i < size_; ++i) {
            out_keys[i] = keys_[perm[i]];
            if (out_values) out_values[i] = vals_[perm[i]];
        }
    }
    void sort_keys(Key* out_keys) const { sort(out_keys, static_cast<Value*>(nullptr)); }

private:
    std::vector<Key> keys_;
    std::vector<Value> vals_;
    int cap_;
    int size_;
};

}  // namespace qadc
