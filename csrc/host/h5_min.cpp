// h5_min: the subset of HDF5 that Keras 2.2.4 `save_weights` files use, with no libhdf5.
//
// Reads and writes: superblock v0, "old style" groups (v1 B-tree + local heap + symbol-table
// nodes), v1 object headers (with continuation blocks), contiguous/compact little-endian float32
// datasets, attributes holding fixed-length string arrays / scalars and variable-length strings
// (global heap).  That is exactly what /root/reference/models/model*/weights.h5 contain (written
// by h5py under Keras 2.2.4, /root/reference/DDL_Normal4.py:172); h5py is not installable in the
// build environment, hence this native replacement of the HDF5 C library.
//
// C ABI (used through ctypes from utils/h5.py); all functions return < 0 on error and store a
// message retrievable with h5min_last_error().
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

#define H5_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

const uint64_t UNDEF = ~0ull;

// ============================================================================ reader
struct Attr {
    std::vector<std::string> strings;   // decoded string payload (fixed or vlen)
    std::vector<float> floats;
    bool is_string = false;
    std::vector<uint64_t> dims;
};

struct Node {
    bool is_group = false;
    std::map<std::string, std::unique_ptr<Node>> children;
    std::vector<std::string> order;             // children in B-tree (name) order
    std::map<std::string, Attr> attrs;
    // dataset
    std::vector<uint64_t> dims;
    int dtype_class = -1, dtype_size = 0;
    uint64_t data_addr = UNDEF, data_size = 0;
    std::vector<uint8_t> compact;
};

// Every byte of the file is untrusted (`load_model` is fed external checkpoints): all reads go through the checked
// accessors below, which throw ParseError; the C entry points catch it (and std::bad_alloc) and return an error.
struct ParseError { std::string msg; };
[[noreturn]] void bad(const std::string& m) { throw ParseError{m}; }

struct File {
    std::vector<uint8_t> buf;
    std::unique_ptr<Node> root;
    int so = 8, sl = 8;
    size_t objects = 0;                      // objects loaded so far (bounded: a crafted file cannot make us spin)
    std::vector<uint64_t> tree_path;         // B-tree nodes on the current descent (cycle guard)

    bool ok(uint64_t off, uint64_t n) const { return off <= buf.size() && n <= buf.size() - off; }
    void need(uint64_t off, uint64_t n, const char* what) const {
        if (!ok(off, n)) bad(std::string("truncated or corrupt file: ") + what);
    }
    uint8_t at(uint64_t off) const { need(off, 1, "byte"); return buf[off]; }
    const uint8_t* ptr(uint64_t off, uint64_t n, const char* what = "range") const { need(off, n, what); return buf.data() + off; }
    uint64_t rd(uint64_t off, int n) const {
        need(off, (uint64_t)n, "field");
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v |= (uint64_t)buf[off + i] << (8 * i);
        return v;
    }
};

const size_t MAX_OBJECTS = 1 << 20;
const int MAX_RANK = 8;

struct Msg { uint16_t type; uint64_t off; uint16_t size; };

bool read_messages(const File& f, uint64_t addr, std::vector<Msg>& out) {
    f.need(addr, 16, "object header");
    int version = f.at(addr);
    if (version != 1) bad("only v1 object headers are supported");
    int nmsg = (int)f.rd(addr + 2, 2);
    uint64_t hsize = f.rd(addr + 8, 4);
    std::vector<std::pair<uint64_t, uint64_t>> blocks{{addr + 16, hsize}};
    for (size_t b = 0; b < blocks.size() && (int)out.size() < nmsg; ++b) {
        if (b > 4096) bad("too many object header continuation blocks");
        uint64_t p = blocks[b].first;
        f.need(p, blocks[b].second, "object header block");
        uint64_t end = p + blocks[b].second;
        while (p + 8 <= end && (int)out.size() < nmsg) {
            uint16_t type = (uint16_t)f.rd(p, 2), size = (uint16_t)f.rd(p + 2, 2);
            f.need(p + 8, size, "header message");
            out.push_back({type, p + 8, size});
            if (type == 0x10) {
                if (size < f.so + f.sl) bad("short continuation message");
                blocks.push_back({f.rd(p + 8, f.so), f.rd(p + 8 + f.so, f.sl)});
            }
            p += 8 + (uint64_t)size;
        }
    }
    return true;
}

std::string heap_string(const File& f, uint64_t heap_addr, uint64_t off) {
    if (memcmp(f.ptr(heap_addr, 32, "local heap"), "HEAP", 4)) bad("bad local heap signature");
    uint64_t seg = f.rd(heap_addr + 8 + 2 * f.sl, f.so);
    if (seg > f.buf.size() || off > f.buf.size() - seg) bad("heap string offset outside the file");
    uint64_t p = seg + off;
    std::string s;
    while (p < f.buf.size() && f.buf[p]) {
        if (s.size() >= 4096) bad("unterminated heap string");
        s.push_back((char)f.buf[p++]);
    }
    return s;
}

struct TypeInfo { int cls = -1, size = 0; bool vlen_string = false; };

TypeInfo parse_type(const File& f, uint64_t off) {
    TypeInfo t;
    f.need(off, 8, "datatype message");
    t.cls = f.at(off) & 0x0f;
    t.size = (int)f.rd(off + 4, 4);
    if (t.size < 0) bad("datatype size out of range");
    if (t.cls == 9) t.vlen_string = (f.at(off + 1) & 0x0f) == 1;
    return t;
}

void parse_space(const File& f, uint64_t off, std::vector<uint64_t>& dims) {
    int version = f.at(off), rank = f.at(off + 1);
    if (rank > MAX_RANK) bad("dataspace rank > 8");
    uint64_t p = off + (version == 1 ? 8 : 4);
    dims.clear();
    for (int i = 0; i < rank; ++i) dims.push_back(f.rd(p + 8 * (uint64_t)i, f.sl));
}

// product of dims, refusing anything that could not possibly be stored in a file of this size
uint64_t checked_count(const File& f, const std::vector<uint64_t>& dims, uint64_t elem_size) {
    uint64_t count = 1;
    const uint64_t limit = f.buf.size() / (elem_size ? elem_size : 1) + 1;
    for (auto d : dims) {
        if (d != 0 && count > limit / d + 1) bad("dataspace larger than the file");
        count *= d;
        if (count > limit && d != 0) bad("dataspace larger than the file");
    }
    return count;
}

uint64_t pad8(uint64_t n) { return (n + 7) & ~7ull; }

bool parse_attr(const File& f, const Msg& m, std::string& name, Attr& a) {
    uint64_t p = m.off;
    int version = f.at(p);
    uint64_t nsz = f.rd(p + 2, 2), tsz = f.rd(p + 4, 2), ssz = f.rd(p + 6, 2);
    p += 8;
    if (version == 3) p += 1;
    auto adv = [&](uint64_t n) { return version == 1 ? pad8(n) : n; };
    const char* np = (const char*)f.ptr(p, nsz, "attribute name");
    name.assign(np, strnlen(np, nsz));
    p += adv(nsz);
    TypeInfo t = parse_type(f, p);
    p += adv(tsz);
    parse_space(f, p, a.dims);
    p += adv(ssz);
    uint64_t elem = t.cls == 9 ? (uint64_t)(4 + f.so + 4) : (uint64_t)(t.size > 0 ? t.size : 1);
    uint64_t count = checked_count(f, a.dims, elem);
    if (count == 0) {
        a.is_string = true;           // Keras writes `weight_names = []` as an empty (float) array
    } else if (t.cls == 3) {
        a.is_string = true;
        if (t.size <= 0) bad("zero-sized string type");
        const char* base = (const char*)f.ptr(p, count * (uint64_t)t.size, "string attribute data");
        for (uint64_t i = 0; i < count; ++i) {
            const char* s = base + i * t.size;
            a.strings.emplace_back(s, strnlen(s, t.size));
        }
    } else if (t.cls == 9 && t.vlen_string) {
        a.is_string = true;
        f.need(p, count * elem, "vlen attribute data");
        for (uint64_t i = 0; i < count; ++i) {
            uint64_t q = p + i * elem;
            uint64_t len = f.rd(q, 4), col = f.rd(q + 4, f.so), idx = f.rd(q + 4 + f.so, 4);
            std::string s;
            if (f.ok(col, 16) && !memcmp(&f.buf[col], "GCOL", 4)) {
                uint64_t csize = f.rd(col + 8, f.sl), o = col + 16;
                if (!f.ok(col, csize)) bad("global heap collection outside the file");
                while (o + 16 <= col + csize) {
                    uint64_t oi = f.rd(o, 2), osz = f.rd(o + 8, f.sl);
                    if (osz > csize) bad("global heap object larger than its collection");
                    if (oi == idx) {
                        uint64_t n = std::min(len, osz);
                        s.assign((const char*)f.ptr(o + 16, n, "global heap object"), (size_t)n);
                        break;
                    }
                    if (oi == 0) break;
                    o += 16 + pad8(osz);
                }
            }
            a.strings.push_back(s);
        }
    } else if (t.cls == 1 && t.size == 4) {
        const uint8_t* src = f.ptr(p, count * 4, "float attribute data");
        a.floats.resize(count);
        memcpy(a.floats.data(), src, count * 4);
    }
    return true;
}

bool load_object(File& f, uint64_t addr, Node& node, int depth);

bool walk_btree(File& f, uint64_t tree, uint64_t heap, Node& group, int depth) {
    if (memcmp(f.ptr(tree, 24, "B-tree node"), "TREE", 4)) bad("bad B-tree node");
    if (f.tree_path.size() > 64 || std::find(f.tree_path.begin(), f.tree_path.end(), tree) != f.tree_path.end())
        bad("B-tree too deep or cyclic");
    f.tree_path.push_back(tree);
    int level = f.at(tree + 5);
    int used = (int)f.rd(tree + 6, 2);
    uint64_t p = tree + 8 + 2 * (uint64_t)f.so;
    for (int i = 0; i < used; ++i) {
        uint64_t child = f.rd(p + f.sl + (uint64_t)i * (f.sl + f.so), f.so);
        if (level > 0) {
            if (!walk_btree(f, child, heap, group, depth)) return false;
            continue;
        }
        if (memcmp(f.ptr(child, 8, "symbol node"), "SNOD", 4)) bad("bad symbol node");
        int n = (int)f.rd(child + 6, 2);
        for (int e = 0; e < n; ++e) {
            uint64_t q = child + 8 + (uint64_t)e * (2 * f.so + 24);
            std::string name = heap_string(f, heap, f.rd(q, f.so));
            uint64_t oh = f.rd(q + f.so, f.so);
            auto c = std::make_unique<Node>();
            if (!load_object(f, oh, *c, depth + 1)) return false;
            if (!group.children.count(name)) group.order.push_back(name);
            group.children[name] = std::move(c);
        }
    }
    f.tree_path.pop_back();
    return true;
}

bool load_object(File& f, uint64_t addr, Node& node, int depth) {
    if (depth > 64) bad("group nesting too deep");
    if (++f.objects > MAX_OBJECTS) bad("too many objects (cyclic or hostile file)");
    std::vector<Msg> msgs;
    read_messages(f, addr, msgs);
    for (auto& m : msgs) {
        if (m.type == 0x11) {
            node.is_group = true;
            if (m.size < 2 * f.so) bad("short symbol table message");
            if (!walk_btree(f, f.rd(m.off, f.so), f.rd(m.off + f.so, f.so), node, depth)) return false;
        } else if (m.type == 0x01) {
            parse_space(f, m.off, node.dims);
        } else if (m.type == 0x03) {
            TypeInfo t = parse_type(f, m.off);
            node.dtype_class = t.cls;
            node.dtype_size = t.size;
        } else if (m.type == 0x08) {
            int version = f.at(m.off);
            if (version != 3) bad("only v3 data layout messages are supported");
            int cls = f.at(m.off + 1);
            if (cls == 1) {
                node.data_addr = f.rd(m.off + 2, f.so);
                node.data_size = f.rd(m.off + 2 + f.so, f.sl);
            } else if (cls == 0) {
                uint64_t n = f.rd(m.off + 2, 2);
                const uint8_t* src = f.ptr(m.off + 4, n, "compact dataset");
                node.compact.assign(src, src + n);
                node.data_size = n;
            } else bad("chunked datasets are not supported");
        } else if (m.type == 0x0c) {
            std::string name;
            Attr a;
            if (parse_attr(f, m, name, a)) node.attrs[name] = std::move(a);
        }
    }
    return true;
}

Node* find(File& f, const char* path) {
    Node* n = f.root.get();
    std::string p(path ? path : "");
    size_t i = 0;
    while (i < p.size()) {
        while (i < p.size() && p[i] == '/') ++i;
        size_t j = p.find('/', i);
        if (j == std::string::npos) j = p.size();
        if (j > i) {
            auto it = n->children.find(p.substr(i, j - i));
            if (it == n->children.end()) return nullptr;
            n = it->second.get();
        }
        i = j;
    }
    return n;
}

int copy_out(const std::string& s, char* buf, int buflen) {
    if (buf && (int)s.size() < buflen) { memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; }
    return (int)s.size();
}

// ============================================================================ writer
struct WAttr { std::string name; std::vector<std::string> strings; bool scalar = false; };
struct WNode {
    bool is_group = true;
    std::map<std::string, std::unique_ptr<WNode>> children;      // std::map == strcmp order for ASCII names
    std::vector<WAttr> attrs;
    std::vector<uint64_t> dims;
    std::vector<float> data;
    // layout
    uint64_t ohdr = 0, btree = 0, heap = 0, heapdata = 0, snod = 0, raw = 0;
};
struct Writer { std::string path; WNode root; };

int LEAF_K = 4;                // group leaf K (a symbol node holds 2K links); raised per file to fit the widest group
const int INTERNAL_K = 16;

struct Out {
    std::vector<uint8_t> b;
    void put(uint64_t off, uint64_t v, int n) {
        if (b.size() < off + n) b.resize(off + n, 0);
        for (int i = 0; i < n; ++i) b[off + i] = (uint8_t)(v >> (8 * i));
    }
    void bytes(uint64_t off, const void* p, size_t n) {
        if (b.size() < off + n) b.resize(off + n, 0);
        memcpy(&b[off], p, n);
    }
};

std::vector<uint8_t> attr_message(const WAttr& a) {
    size_t maxlen = 1;
    for (auto& s : a.strings) maxlen = std::max(maxlen, s.size());
    std::vector<uint8_t> m;
    auto p16 = [&](uint16_t v) { m.push_back(v & 0xff); m.push_back(v >> 8); };
    std::string nm = a.name + std::string(1, '\0');
    size_t nsz = nm.size();
    int rank = a.scalar ? 0 : 1;
    size_t ssz = 8 + 8 * rank;
    m.push_back(1); m.push_back(0);
    p16((uint16_t)nsz); p16(8); p16((uint16_t)ssz);
    m.insert(m.end(), nm.begin(), nm.end());
    m.resize(8 + pad8(nsz), 0);
    // datatype: class 3 (string) version 1, null-padded ASCII, size = maxlen
    uint8_t dt[8] = {0x13, 0x01, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) dt[4 + i] = (uint8_t)(maxlen >> (8 * i));
    m.insert(m.end(), dt, dt + 8);
    // dataspace v1
    uint8_t sp[8] = {1, (uint8_t)rank, 0, 0, 0, 0, 0, 0};
    m.insert(m.end(), sp, sp + 8);
    if (rank) { uint64_t n = a.strings.size(); for (int i = 0; i < 8; ++i) m.push_back((uint8_t)(n >> (8 * i))); }
    for (auto& s : a.strings) { size_t o = m.size(); m.resize(o + maxlen, 0); memcpy(&m[o], s.data(), s.size()); }
    m.resize(pad8(m.size()), 0);
    return m;
}

struct MsgOut { uint16_t type; std::vector<uint8_t> data; };

std::vector<MsgOut> node_messages(const WNode& n) {
    std::vector<MsgOut> ms;
    if (n.is_group) {
        std::vector<uint8_t> d(16, 0);
        ms.push_back({0x11, d});      // patched later with btree/heap addresses
    } else {
        int rank = (int)n.dims.size();
        std::vector<uint8_t> sp(8 + 8 * rank, 0);
        sp[0] = 1; sp[1] = (uint8_t)rank;
        for (int i = 0; i < rank; ++i) for (int k = 0; k < 8; ++k) sp[8 + 8 * i + k] = (uint8_t)(n.dims[i] >> (8 * k));
        ms.push_back({0x01, sp});
        // IEEE little-endian float32: class 1 v1, bits {0x20, 0x1f, 0x00}, size 4,
        // properties: bit offset 0, precision 32, exp loc 23, exp size 8, mant loc 0, mant size 23, bias 127
        std::vector<uint8_t> dt = {0x11, 0x20, 0x1f, 0x00, 4, 0, 0, 0, 0, 0, 32, 0, 23, 8, 0, 23, 127, 0, 0, 0};
        dt.resize(24, 0);
        ms.push_back({0x03, dt});
        std::vector<uint8_t> fv = {2, 2, 2, 0};     // fill value v2: alloc late, write if-set, undefined
        fv.resize(8, 0);
        ms.push_back({0x05, fv});
        std::vector<uint8_t> lay(24, 0);
        lay[0] = 3; lay[1] = 1;                      // v3 contiguous; address/size patched later
        ms.push_back({0x08, lay});
    }
    for (auto& a : n.attrs) ms.push_back({0x0c, attr_message(a)});
    return ms;
}

uint64_t layout(WNode& n, uint64_t at) {
    auto ms = node_messages(n);
    uint64_t hsize = 0;
    for (auto& m : ms) hsize += 8 + m.data.size();
    n.ohdr = at;
    at += 16 + hsize;
    if (n.is_group) {
        n.btree = at; at += 8 + 16 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8;
        n.heap = at; at += 32;
        uint64_t hd = 8;
        for (auto& c : n.children) hd += pad8(c.first.size() + 1);
        n.heapdata = at; at += pad8(std::max<uint64_t>(hd, 16) + 16);
        n.snod = at; at += 8 + (uint64_t)2 * LEAF_K * 40;
        for (auto& c : n.children) at = layout(*c.second, at);
    } else {
        n.raw = at; at += pad8(n.data.size() * 4);
    }
    return at;
}

void emit(const WNode& n, Out& o) {
    auto ms = node_messages(n);
    uint64_t hsize = 0;
    for (auto& m : ms) hsize += 8 + m.data.size();
    o.put(n.ohdr, 1, 1);
    o.put(n.ohdr + 2, ms.size(), 2);
    o.put(n.ohdr + 4, 1, 4);
    o.put(n.ohdr + 8, hsize, 4);
    uint64_t p = n.ohdr + 16;
    for (auto& m : ms) {
        if (m.type == 0x11) { for (int k = 0; k < 8; ++k) { m.data[k] = (uint8_t)(n.btree >> (8 * k)); m.data[8 + k] = (uint8_t)(n.heap >> (8 * k)); } }
        if (m.type == 0x08) {
            uint64_t sz = n.data.size() * 4, addr = sz ? n.raw : UNDEF;
            for (int k = 0; k < 8; ++k) { m.data[2 + k] = (uint8_t)(addr >> (8 * k)); m.data[10 + k] = (uint8_t)(sz >> (8 * k)); }
        }
        o.put(p, m.type, 2);
        o.put(p + 2, m.data.size(), 2);
        o.put(p + 4, m.type == 0x05 ? 1 : 0, 1);
        o.bytes(p + 8, m.data.data(), m.data.size());
        p += 8 + m.data.size();
    }
    if (!n.is_group) {
        if (!n.data.empty()) o.bytes(n.raw, n.data.data(), n.data.size() * 4);
        return;
    }
    // local heap: offset 0 = "" ; then child names
    std::map<std::string, uint64_t> off;
    uint64_t hd = 8;
    for (auto& c : n.children) { off[c.first] = hd; hd += pad8(c.first.size() + 1); }
    uint64_t seg = pad8(std::max<uint64_t>(hd, 16) + 16);
    o.bytes(n.heap, "HEAP", 4);
    o.put(n.heap + 8, seg, 8);
    o.put(n.heap + 16, hd, 8);                 // free-list head: one free block after the names
    o.put(n.heap + 24, n.heapdata, 8);
    for (auto& c : n.children) o.bytes(n.heapdata + off[c.first], c.first.c_str(), c.first.size() + 1);
    o.put(n.heapdata + hd, 1, 8);              // free block: next = 1 (none), size = rest
    o.put(n.heapdata + hd + 8, seg - hd, 8);
    // B-tree: single leaf-level node pointing at one symbol node
    o.bytes(n.btree, "TREE", 4);
    o.put(n.btree + 4, 0, 1);
    o.put(n.btree + 5, 0, 1);
    o.put(n.btree + 6, n.children.empty() ? 0 : 1, 2);
    o.put(n.btree + 8, UNDEF, 8);
    o.put(n.btree + 16, UNDEF, 8);
    o.put(n.btree + 24, 0, 8);                                            // key 0: ""
    o.put(n.btree + 32, n.snod, 8);
    o.put(n.btree + 40, n.children.empty() ? 0 : off[n.children.rbegin()->first], 8);   // key 1: largest name
    o.bytes(n.snod, "SNOD", 4);
    o.put(n.snod + 4, 1, 1);
    o.put(n.snod + 6, n.children.size(), 2);
    uint64_t q = n.snod + 8;
    for (auto& c : n.children) {
        o.put(q, off[c.first], 8);
        o.put(q + 8, c.second->ohdr, 8);
        if (c.second->is_group) {
            o.put(q + 16, 1, 4);
            o.put(q + 24, c.second->btree, 8);
            o.put(q + 32, c.second->heap, 8);
        }
        q += 40;
    }
    for (auto& c : n.children) emit(*c.second, o);
}

WNode* wfind(Writer& w, const char* path, bool create_groups) {
    WNode* n = &w.root;
    std::string p(path ? path : "");
    size_t i = 0;
    while (i < p.size()) {
        while (i < p.size() && p[i] == '/') ++i;
        size_t j = p.find('/', i);
        if (j == std::string::npos) j = p.size();
        if (j > i) {
            std::string part = p.substr(i, j - i);
            auto it = n->children.find(part);
            if (it == n->children.end()) {
                if (!create_groups) return nullptr;
                it = n->children.emplace(part, std::make_unique<WNode>()).first;
            }
            n = it->second.get();
        }
        i = j;
    }
    return n;
}

}  // namespace

// ============================================================================ C ABI
H5_API const char* h5min_last_error() { return g_err.c_str(); }

H5_API void* h5min_open(const char* path) {
    g_err.clear();
    try {
        FILE* fp = fopen(path, "rb");
        if (!fp) { fail(std::string("cannot open ") + path); return nullptr; }
        auto f = std::make_unique<File>();
        fseek(fp, 0, SEEK_END);
        long n = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        f->buf.resize(n > 0 ? n : 0);
        size_t got = fread(f->buf.data(), 1, f->buf.size(), fp);
        fclose(fp);
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        if (got != f->buf.size() || f->buf.size() < 96 || memcmp(f->buf.data(), sig, 8)) { fail("not an HDF5 file"); return nullptr; }
        if (f->buf[8] != 0) { fail("only superblock version 0 is supported"); return nullptr; }
        f->so = f->buf[13];
        f->sl = f->buf[14];
        if (f->so != 8 || f->sl != 8) { fail("only 8-byte offsets/lengths are supported"); return nullptr; }
        uint64_t root_ohdr = f->rd(56 + 8, 8);
        f->root = std::make_unique<Node>();
        if (!load_object(*f, root_ohdr, *f->root, 0)) { if (g_err.empty()) fail("corrupt file"); return nullptr; }
        return f.release();
    } catch (const ParseError& e) {
        fail(e.msg);
    } catch (const std::exception& e) {       // bad_alloc / length_error from a hostile size field
        fail(std::string("corrupt file: ") + e.what());
    }
    return nullptr;
}

H5_API void h5min_close(void* h) { delete (File*)h; }

// children of a group, '\n'-separated; returns the needed length
H5_API int h5min_list(void* h, const char* path, char* buf, int buflen) {
    Node* n = find(*(File*)h, path);
    if (!n || !n->is_group) return fail("no such group");
    std::string s;
    for (auto& c : n->order) { if (!s.empty()) s += '\n'; s += c; }
    return copy_out(s, buf, buflen);
}

H5_API int h5min_is_group(void* h, const char* path) {
    Node* n = find(*(File*)h, path);
    if (!n) return fail("no such object");
    return n->is_group ? 1 : 0;
}

// names of the attributes of an object, '\n'-separated
H5_API int h5min_attr_names(void* h, const char* path, char* buf, int buflen) {
    Node* n = find(*(File*)h, path);
    if (!n) return fail("no such object");
    std::string s;
    for (auto& a : n->attrs) { if (!s.empty()) s += '\n'; s += a.first; }
    return copy_out(s, buf, buflen);
}

// string attribute (array elements '\n'-separated)
H5_API int h5min_attr_strings(void* h, const char* path, const char* name, char* buf, int buflen) {
    Node* n = find(*(File*)h, path);
    if (!n) return fail("no such object");
    auto it = n->attrs.find(name);
    if (it == n->attrs.end() || !it->second.is_string) return fail("no such string attribute");
    std::string s;
    for (size_t i = 0; i < it->second.strings.size(); ++i) { if (i) s += '\n'; s += it->second.strings[i]; }
    return copy_out(s, buf, buflen);
}

// dataset shape: returns rank, fills dims[<=8]; *dtype = 1 for float32 else 0
H5_API int h5min_dataset_info(void* h, const char* path, uint64_t* dims, int* dtype) {
    Node* n = find(*(File*)h, path);
    if (!n || n->is_group) return fail("no such dataset");
    if (n->dims.size() > 8) return fail("rank > 8");
    for (size_t i = 0; i < n->dims.size(); ++i) dims[i] = n->dims[i];
    *dtype = (n->dtype_class == 1 && n->dtype_size == 4) ? 1 : 0;
    return (int)n->dims.size();
}

H5_API int h5min_read_f32(void* h, const char* path, float* out, uint64_t count) {
    File& f = *(File*)h;
    Node* n = find(f, path);
    if (!n || n->is_group) return fail("no such dataset");
    if (!(n->dtype_class == 1 && n->dtype_size == 4)) return fail("dataset is not float32");
    uint64_t total = 1;
    for (auto d : n->dims) {
        if (d != 0 && total > (f.buf.size() / 4 + 1) / d + 1) return fail("dataset larger than the file");
        total *= d;
    }
    if (total != count) return fail("element count mismatch");
    if (total == 0) return 0;
    if (total > f.buf.size() / 4 + 1) return fail("dataset larger than the file");
    if (!n->compact.empty()) {
        if (n->compact.size() < total * 4) return fail("compact data truncated");
        memcpy(out, n->compact.data(), total * 4);
        return 0;
    }
    if (n->data_addr == UNDEF || !f.ok(n->data_addr, total * 4)) return fail("dataset storage missing/truncated");
    memcpy(out, &f.buf[n->data_addr], total * 4);
    return 0;
}

H5_API void* h5min_create(const char* path) {
    g_err.clear();
    auto w = new Writer();
    w->path = path;
    return w;
}

H5_API int h5min_add_group(void* h, const char* path) {
    WNode* n = wfind(*(Writer*)h, path, true);
    return n && n->is_group ? 0 : fail("path exists and is not a group");
}

// strings: '\n'-separated; scalar != 0 writes a rank-0 attribute
H5_API int h5min_add_attr_strings(void* h, const char* path, const char* name, const char* strings, int scalar) {
    WNode* n = wfind(*(Writer*)h, path, false);
    if (!n) return fail("no such object");
    WAttr a;
    a.name = name;
    a.scalar = scalar != 0;
    std::string s(strings ? strings : "");
    if (a.scalar) a.strings.push_back(s);
    else if (!s.empty()) {
        size_t i = 0;
        while (i <= s.size()) {
            size_t j = s.find('\n', i);
            if (j == std::string::npos) j = s.size();
            a.strings.push_back(s.substr(i, j - i));
            i = j + 1;
        }
    }
    n->attrs.push_back(std::move(a));
    return 0;
}

H5_API int h5min_add_dataset_f32(void* h, const char* path, const uint64_t* dims, int rank, const float* data) {
    Writer& w = *(Writer*)h;
    std::string p(path);
    size_t k = p.find_last_of('/');
    std::string parent = k == std::string::npos ? "" : p.substr(0, k), leaf = k == std::string::npos ? p : p.substr(k + 1);
    WNode* g = wfind(w, parent.c_str(), true);
    if (!g || !g->is_group || leaf.empty()) return fail("bad dataset path");
    if (g->children.count(leaf)) return fail("object already exists");
    auto d = std::make_unique<WNode>();
    d->is_group = false;
    uint64_t total = 1;
    for (int i = 0; i < rank; ++i) { d->dims.push_back(dims[i]); total *= dims[i]; }
    d->data.assign(data, data + total);
    g->children[leaf] = std::move(d);
    return 0;
}

// serialise and close
H5_API int h5min_finalize(void* h) {
    std::unique_ptr<Writer> w((Writer*)h);
    struct R { static size_t widest(const WNode& n) { size_t m = n.children.size(); for (auto& c : n.children) m = std::max(m, widest(*c.second)); return m; } };
    size_t widest = R::widest(w->root);
    if (widest > 60000) return fail("too many links in one group");
    LEAF_K = (int)std::max<size_t>(4, (widest + 1) / 2);
    uint64_t eof = layout(w->root, 96);
    Out o;
    o.b.assign(eof, 0);
    static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    o.bytes(0, sig, 8);
    o.put(13, 8, 1);
    o.put(14, 8, 1);
    o.put(16, LEAF_K, 2);
    o.put(18, INTERNAL_K, 2);
    o.put(24, 0, 8);
    o.put(32, UNDEF, 8);
    o.put(40, eof, 8);
    o.put(48, UNDEF, 8);
    o.put(56, 0, 8);
    o.put(64, w->root.ohdr, 8);
    o.put(72, 1, 4);
    o.put(80, w->root.btree, 8);
    o.put(88, w->root.heap, 8);
    emit(w->root, o);
    o.b.resize(eof, 0);
    FILE* fp = fopen(w->path.c_str(), "wb");
    if (!fp) return fail("cannot create " + w->path);
    size_t put = fwrite(o.b.data(), 1, o.b.size(), fp);
    fclose(fp);
    return put == o.b.size() ? 0 : fail("short write");
}
