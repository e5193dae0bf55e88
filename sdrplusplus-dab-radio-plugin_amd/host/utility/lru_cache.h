// utility/lru_cache.h -- least-recently-used map with the interface the GUI uses for its texture cache
// (/root/reference/src/render_radio_block.h:9, :23-26): set_max_size(), find(), emplace().
#pragma once
#include <cstddef>
#include <list>
#include <unordered_map>
#include <utility>

template <class K, class V>
class LRU_Cache {
public:
    void set_max_size(size_t n) { m_max = n; trim(); }
    size_t get_max_size() const { return m_max; }
    size_t size() const { return m_items.size(); }
    V *find(const K &key) {
        auto it = m_index.find(key);
        if (it == m_index.end()) return nullptr;
        m_items.splice(m_items.begin(), m_items, it->second);        // most recently used first
        return &it->second->second;
    }
    template <class... Args>
    V &emplace(const K &key, Args &&...args) {
        if (V *v = find(key)) return *v;
        m_items.emplace_front(std::piecewise_construct, std::forward_as_tuple(key), std::forward_as_tuple(std::forward<Args>(args)...));
        m_index[key] = m_items.begin();
        trim();
        return m_items.front().second;
    }
    void remove(const K &key) {
        auto it = m_index.find(key);
        if (it == m_index.end()) return;
        m_items.erase(it->second);
        m_index.erase(it);
    }

private:
    void trim() {
        while (m_items.size() > m_max) {
            m_index.erase(m_items.back().first);
            m_items.pop_back();
        }
    }
    size_t m_max = 64;
    std::list<std::pair<K, V>> m_items;
    std::unordered_map<K, typename std::list<std::pair<K, V>>::iterator> m_index;
};
