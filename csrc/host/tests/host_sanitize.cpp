// Sanitizer driver for the host runtime (built with -fsanitize=address,undefined or =thread together with
// h5_min.cpp and actor_transport.cpp by `python build.py --sanitize asan|ubsan|tsan`; CPU only).
//
//   host_sanitize h5fuzz <iterations> <file.h5>...   mutation fuzz of the HDF5 reader: truncations, bit flips, byte
//                                                    runs and hostile 8-byte fields over real Keras weight files;
//                                                    every mutant must be rejected with an error or read cleanly
//   host_sanitize transport <threads> <messages>     N client threads hammer one listening endpoint with frames
//                                                    while the driver echoes them: races in the epoll thread / send
//                                                    mutexes / event queue show up under TSan, lifetime bugs under ASan
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

extern "C" {
const char* h5min_last_error();
void* h5min_open(const char* path);
void h5min_close(void* h);
int h5min_list(void* h, const char* path, char* buf, int buflen);
int h5min_is_group(void* h, const char* path);
int h5min_attr_names(void* h, const char* path, char* buf, int buflen);
int h5min_attr_strings(void* h, const char* path, const char* name, char* buf, int buflen);
int h5min_dataset_info(void* h, const char* path, uint64_t* dims, int* dtype);
int h5min_read_f32(void* h, const char* path, float* out, uint64_t count);

const char* ddl_ep_last_error();
void* ddl_ep_create();
int ddl_ep_listen(void* h, const char* path);
int ddl_ep_connect(void* h, const char* path, int timeout_ms);
int ddl_ep_send(void* h, int conn, uint64_t tag, const void* buf, uint64_t n);
int ddl_ep_next(void* h, int timeout_ms, int* conn, uint64_t* tag, uint64_t* size);
int ddl_ep_take(void* h, void* buf, uint64_t cap);
int ddl_ep_num_conns(void* h);
void ddl_ep_close_conn(void* h, int conn);
void ddl_ep_destroy(void* h);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

static std::vector<std::string> split(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return out;
}

// Walk everything the Python loader would touch.  Returns the number of objects visited.
static long walk(void* h, const std::string& path, int depth) {
    if (depth > 16) return 0;
    long seen = 1;
    std::vector<char> buf(1 << 16);
    int n = h5min_attr_names(h, path.c_str(), buf.data(), (int)buf.size());
    if (n > 0 && n < (int)buf.size()) {
        for (auto& a : split(std::string(buf.data(), n))) {
            std::vector<char> v(1 << 16);
            h5min_attr_strings(h, path.c_str(), a.c_str(), v.data(), (int)v.size());
        }
    }
    int g = h5min_is_group(h, path.c_str());
    if (g == 1) {
        n = h5min_list(h, path.c_str(), buf.data(), (int)buf.size());
        if (n > 0 && n < (int)buf.size())
            for (auto& c : split(std::string(buf.data(), n))) seen += walk(h, path + "/" + c, depth + 1);
    } else if (g == 0) {
        uint64_t dims[8];
        int dtype = 0;
        int rank = h5min_dataset_info(h, path.c_str(), dims, &dtype);
        if (rank >= 0 && dtype == 1) {
            uint64_t total = 1;
            bool sane = true;
            for (int i = 0; i < rank; ++i) {
                if (dims[i] > (1ull << 28) || total > (1ull << 28)) { sane = false; break; }
                total *= dims[i];
            }
            if (sane && total <= (1ull << 26)) {
                std::vector<float> out(total ? total : 1);
                h5min_read_f32(h, path.c_str(), out.data(), total);
            }
        }
    }
    return seen;
}

static int h5fuzz(int iters, int nfiles, char** files) {
    char tmpl[] = "/tmp/h5fuzzXXXXXX";
    int fd = mkstemp(tmpl);
    if (fd < 0) { perror("mkstemp"); return 2; }
    close(fd);
    long opened = 0, rejected = 0, visited = 0;
    for (int fi = 0; fi < nfiles; ++fi) {
        FILE* fp = fopen(files[fi], "rb");
        if (!fp) { fprintf(stderr, "cannot read %s\n", files[fi]); return 2; }
        std::vector<uint8_t> orig;
        uint8_t chunk[65536];
        size_t k;
        while ((k = fread(chunk, 1, sizeof(chunk), fp)) > 0) orig.insert(orig.end(), chunk, chunk + k);
        fclose(fp);
        // metadata of these files sits in the first few KB and around object headers; bias mutations there
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = orig;
            int kind = it == 0 ? -1 : (int)(rnd() % 5);
            auto pos = [&]() -> size_t {
                uint64_t r = rnd();
                size_t lim = (r & 3) ? std::min<size_t>(m.size(), 8192) : m.size();
                return (size_t)((r >> 8) % lim);
            };
            if (kind == 0) m.resize((size_t)(rnd() % m.size()));                                    // truncate
            else if (kind == 1) { int nf = 1 + (int)(rnd() % 8); for (int j = 0; j < nf; ++j) m[pos()] ^= (uint8_t)(1u << (rnd() % 8)); }
            else if (kind == 2) { size_t p = pos(), len = 1 + (size_t)(rnd() % 32); for (size_t j = p; j < m.size() && j < p + len; ++j) m[j] = (uint8_t)rnd(); }
            else if (kind == 3) {                                                                   // hostile 8-byte field
                static const uint64_t vals[] = {0, 1, 0xFFFFFFFFFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFull, 0x100000000ull};
                uint64_t v = (rnd() & 1) ? vals[rnd() % 6] : (uint64_t)pos();                        // or a self/loop pointer
                size_t p = pos() & ~(size_t)7;
                if (p + 8 <= m.size()) memcpy(&m[p], &v, 8);
            } else if (kind == 4) { size_t p = pos(); uint16_t v = (uint16_t)rnd(); if (p + 2 <= m.size()) memcpy(&m[p], &v, 2); }
            fp = fopen(tmpl, "wb");
            fwrite(m.data(), 1, m.size(), fp);
            fclose(fp);
            void* h = h5min_open(tmpl);
            if (!h) { ++rejected; if (!h5min_last_error()[0]) { fprintf(stderr, "rejected without a message\n"); return 1; } continue; }
            ++opened;
            visited += walk(h, "", 0);
            h5min_close(h);
        }
    }
    unlink(tmpl);
    printf("h5fuzz: %ld mutants opened and walked (%ld objects), %ld rejected with an error, 0 crashes\n", opened, visited, rejected);
    return 0;
}

static int transport(int nthreads, int nmsg) {
    std::string path = "/tmp/ddl_san_" + std::to_string(getpid()) + ".sock";
    void* srv = ddl_ep_create();
    if (!srv || ddl_ep_listen(srv, path.c_str()) != 0) { fprintf(stderr, "listen: %s\n", ddl_ep_last_error()); return 2; }
    std::atomic<long> echoed{0}, bad{0};
    std::vector<std::thread> clients;
    for (int t = 0; t < nthreads; ++t) {
        clients.emplace_back([&, t] {
            void* ep = ddl_ep_create();
            int c = ddl_ep_connect(ep, path.c_str(), 5000);
            if (c < 0) { ++bad; ddl_ep_destroy(ep); return; }
            std::vector<char> payload(1 + (t * 977) % 5000, (char)t), back(8192);
            // a second thread of the same client sends concurrently on the same connection (send mutex)
            std::thread side([&] { for (int i = 0; i < nmsg; ++i) ddl_ep_send(ep, c, 1000 + t, payload.data(), payload.size()); });
            for (int i = 0; i < nmsg; ++i) ddl_ep_send(ep, c, t, payload.data(), payload.size());
            side.join();
            long got = 0;
            while (got < 2L * nmsg) {
                int conn; uint64_t tag, size;
                int k = ddl_ep_next(ep, 5000, &conn, &tag, &size);
                if (k == 2) continue;                          // 2 = connected
                if (k != 1) { ++bad; break; }                  // 1 = message
                if (size > back.size()) back.resize(size);
                ddl_ep_take(ep, back.data(), back.size());
                if (size != payload.size() || back[0] != (char)t) ++bad;
                ++got;
            }
            echoed += got;
            ddl_ep_destroy(ep);
        });
    }
    long total = 2L * nthreads * nmsg, served = 0;
    std::vector<char> buf(8192);
    int closed = 0;
    while (served < total && closed < nthreads) {
        int conn; uint64_t tag, size;
        int k = ddl_ep_next(srv, 10000, &conn, &tag, &size);
        if (k == 0) break;                                      // timeout
        if (k == 1) {
            if (size > buf.size()) buf.resize(size);
            ddl_ep_take(srv, buf.data(), buf.size());
            ddl_ep_send(srv, conn, tag, buf.data(), size);
            ++served;
        } else if (k == 3) ++closed;                            // peer closed
    }
    for (auto& c : clients) c.join();
    ddl_ep_destroy(srv);
    printf("transport: %d threads x 2 senders x %d messages: %ld served, %ld echoed back, %ld errors\n", nthreads, nmsg,
           served, echoed.load(), bad.load());
    return (bad.load() == 0 && echoed.load() == total) ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "h5fuzz")) return h5fuzz(atoi(argv[2]), argc - 3, argv + 3);
    if (argc == 4 && !strcmp(argv[1], "transport")) return transport(atoi(argv[2]), atoi(argv[3]));
    fprintf(stderr, "usage: host_sanitize h5fuzz <iters> <file.h5>... | transport <threads> <messages>\n");
    return 2;
}
