// Actor-runtime transport: framed messages over Unix-domain sockets with a native receive thread.
//
// This is the process-boundary layer of the ensemble runtime (the role Ray's raylet/object-store
// transport played for /root/reference/DDL_Ray*.py): a driver endpoint listens, one
// worker process per GPU connects, both sides exchange length-prefixed frames.  An `Endpoint` owns
// an epoll loop on its own thread that reassembles frames from every connection into a single
// event queue, so the Python side never blocks on a slow peer and a dead worker surfaces as a
// CLOSED event (failure detection) instead of a hang.
//
// Plain C ABI (ctypes-friendly); no Python, no torch.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define EP_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr uint32_t kMagic = 0x44444C41u;   // "DDLA"
constexpr uint64_t kMaxFrame = 1ull << 34; // 16 GiB sanity bound

struct Header {
    uint32_t magic;
    uint32_t reserved;
    uint64_t tag;
    uint64_t size;
};

enum EventKind { EV_TIMEOUT = 0, EV_MESSAGE = 1, EV_CONNECTED = 2, EV_CLOSED = 3 };

struct Event {
    int kind;
    int conn;
    uint64_t tag;
    std::vector<char> payload;
};

struct Conn {
    int fd = -1;
    int id = -1;
    std::mutex send_mu;
    // receive state machine
    Header hdr{};
    size_t hdr_got = 0;
    std::vector<char> body;
    size_t body_got = 0;
    bool in_body = false;
};

struct Endpoint {
    int epfd = -1;
    int wakefd = -1;
    int listenfd = -1;
    std::string listen_path;
    std::thread loop;
    std::atomic<bool> stop{false};
    std::mutex mu;                       // guards conns, queue
    std::condition_variable cv;
    std::map<int, std::shared_ptr<Conn>> conns;
    std::deque<Event> queue;
    Event current;                       // last event handed to the caller
    int next_id = 1;
    std::string err;

    void push(Event&& e) {
        {
            std::lock_guard<std::mutex> g(mu);
            queue.push_back(std::move(e));
        }
        cv.notify_all();
    }
};

thread_local std::string g_err;

void set_nonblock(int fd, bool nb) {
    int fl = fcntl(fd, F_GETFL, 0);
    fcntl(fd, F_SETFL, nb ? (fl | O_NONBLOCK) : (fl & ~O_NONBLOCK));
}

int add_conn(Endpoint* ep, int fd) {
    auto c = std::make_shared<Conn>();
    c->fd = fd;
    set_nonblock(fd, true);
    {
        std::lock_guard<std::mutex> g(ep->mu);
        c->id = ep->next_id++;
        ep->conns[c->id] = c;
    }
    epoll_event ev{};
    ev.events = EPOLLIN | EPOLLRDHUP;
    ev.data.u64 = (uint64_t)c->id;
    epoll_ctl(ep->epfd, EPOLL_CTL_ADD, fd, &ev);
    return c->id;
}

void drop_conn(Endpoint* ep, int id, bool notify) {
    std::shared_ptr<Conn> c;
    {
        std::lock_guard<std::mutex> g(ep->mu);
        auto it = ep->conns.find(id);
        if (it == ep->conns.end()) return;
        c = it->second;
        ep->conns.erase(it);
    }
    epoll_ctl(ep->epfd, EPOLL_CTL_DEL, c->fd, nullptr);
    {
        std::lock_guard<std::mutex> g(c->send_mu);
        close(c->fd);
        c->fd = -1;
    }
    if (notify) ep->push(Event{EV_CLOSED, id, 0, {}});
}

// Drain everything readable on `c`; returns false when the peer is gone.
bool pump(Endpoint* ep, const std::shared_ptr<Conn>& c) {
    for (;;) {
        if (!c->in_body) {
            ssize_t n = read(c->fd, (char*)&c->hdr + c->hdr_got, sizeof(Header) - c->hdr_got);
            if (n == 0) return false;
            if (n < 0) return errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR;
            c->hdr_got += (size_t)n;
            if (c->hdr_got < sizeof(Header)) continue;
            if (c->hdr.magic != kMagic || c->hdr.size > kMaxFrame) return false;   // corrupt stream
            c->body.assign((size_t)c->hdr.size, 0);
            c->body_got = 0;
            c->in_body = true;
        }
        while (c->body_got < c->body.size()) {
            ssize_t n = read(c->fd, c->body.data() + c->body_got, c->body.size() - c->body_got);
            if (n == 0) return false;
            if (n < 0) return errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR;
            c->body_got += (size_t)n;
        }
        ep->push(Event{EV_MESSAGE, c->id, c->hdr.tag, std::move(c->body)});
        c->body = std::vector<char>();
        c->hdr_got = 0;
        c->in_body = false;
    }
}

void loop_fn(Endpoint* ep) {
    epoll_event evs[32];
    while (!ep->stop.load()) {
        int n = epoll_wait(ep->epfd, evs, 32, 200);
        if (n < 0) {
            if (errno == EINTR) continue;
            break;
        }
        for (int i = 0; i < n; ++i) {
            uint64_t key = evs[i].data.u64;
            if (key == 0) {            // wake / stop
                uint64_t v;
                (void)!read(ep->wakefd, &v, sizeof(v));
                continue;
            }
            if (key == UINT64_MAX) {   // listener
                for (;;) {
                    int fd = accept(ep->listenfd, nullptr, nullptr);
                    if (fd < 0) break;
                    int id = add_conn(ep, fd);
                    ep->push(Event{EV_CONNECTED, id, 0, {}});
                }
                continue;
            }
            std::shared_ptr<Conn> c;
            {
                std::lock_guard<std::mutex> g(ep->mu);
                auto it = ep->conns.find((int)key);
                if (it != ep->conns.end()) c = it->second;
            }
            if (!c) continue;
            bool alive = true;
            if (evs[i].events & EPOLLIN) alive = pump(ep, c);
            if (alive && (evs[i].events & (EPOLLHUP | EPOLLERR | EPOLLRDHUP))) alive = pump(ep, c) && false;
            if (!alive) drop_conn(ep, c->id, true);
        }
    }
}

bool write_all(int fd, const char* p, size_t n) {
    while (n) {
        ssize_t w = send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            if (errno == EAGAIN || errno == EWOULDBLOCK) {   // socket buffer full: wait until writable
                fd_set ws;
                FD_ZERO(&ws);
                FD_SET(fd, &ws);
                timeval tv{5, 0};
                select(fd + 1, nullptr, &ws, nullptr, &tv);
                continue;
            }
            return false;
        }
        p += w;
        n -= (size_t)w;
    }
    return true;
}

bool fill_addr(const char* path, sockaddr_un* a, socklen_t* len) {
    memset(a, 0, sizeof(*a));
    a->sun_family = AF_UNIX;
    size_t n = strlen(path);
    if (n == 0 || n >= sizeof(a->sun_path)) return false;
    memcpy(a->sun_path, path, n);
    *len = (socklen_t)(offsetof(sockaddr_un, sun_path) + n + 1);
    return true;
}

}  // namespace

EP_API const char* ddl_ep_last_error() { return g_err.c_str(); }

EP_API void* ddl_ep_create() {
    auto* ep = new Endpoint();
    ep->epfd = epoll_create1(EPOLL_CLOEXEC);
    ep->wakefd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    if (ep->epfd < 0 || ep->wakefd < 0) {
        g_err = std::string("epoll/eventfd: ") + strerror(errno);
        delete ep;
        return nullptr;
    }
    epoll_event ev{};
    ev.events = EPOLLIN;
    ev.data.u64 = 0;
    epoll_ctl(ep->epfd, EPOLL_CTL_ADD, ep->wakefd, &ev);
    ep->loop = std::thread(loop_fn, ep);
    return ep;
}

EP_API int ddl_ep_listen(void* h, const char* path) {
    auto* ep = (Endpoint*)h;
    sockaddr_un a;
    socklen_t len;
    if (!fill_addr(path, &a, &len)) { g_err = "bad socket path"; return -1; }
    int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) { g_err = strerror(errno); return -1; }
    unlink(path);
    if (bind(fd, (sockaddr*)&a, len) < 0 || listen(fd, 128) < 0) {
        g_err = std::string("bind/listen ") + path + ": " + strerror(errno);
        close(fd);
        return -1;
    }
    set_nonblock(fd, true);
    ep->listenfd = fd;
    ep->listen_path = path;
    epoll_event ev{};
    ev.events = EPOLLIN;
    ev.data.u64 = UINT64_MAX;
    epoll_ctl(ep->epfd, EPOLL_CTL_ADD, fd, &ev);
    return 0;
}

EP_API int ddl_ep_connect(void* h, const char* path, int timeout_ms) {
    auto* ep = (Endpoint*)h;
    sockaddr_un a;
    socklen_t len;
    if (!fill_addr(path, &a, &len)) { g_err = "bad socket path"; return -1; }
    auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
        int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
        if (fd < 0) { g_err = strerror(errno); return -1; }
        if (connect(fd, (sockaddr*)&a, len) == 0) return add_conn(ep, fd);
        int e = errno;
        close(fd);
        if (std::chrono::steady_clock::now() >= deadline) {
            g_err = std::string("connect ") + path + ": " + strerror(e);
            return -1;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

EP_API int ddl_ep_send(void* h, int conn, uint64_t tag, const void* buf, uint64_t n) {
    auto* ep = (Endpoint*)h;
    std::shared_ptr<Conn> c;
    {
        std::lock_guard<std::mutex> g(ep->mu);
        auto it = ep->conns.find(conn);
        if (it == ep->conns.end()) { g_err = "connection closed"; return -2; }
        c = it->second;
    }
    Header hd{kMagic, 0, tag, n};
    std::lock_guard<std::mutex> g(c->send_mu);
    if (c->fd < 0) { g_err = "connection closed"; return -2; }
    if (!write_all(c->fd, (const char*)&hd, sizeof(hd)) || !write_all(c->fd, (const char*)buf, (size_t)n)) {
        g_err = std::string("send: ") + strerror(errno);
        return -2;
    }
    return 0;
}

// Blocks up to timeout_ms for the next event.  Returns the event kind; for EV_MESSAGE the payload
// stays owned by the endpoint until the next call and is copied out with ddl_ep_take().
EP_API int ddl_ep_next(void* h, int timeout_ms, int* conn, uint64_t* tag, uint64_t* size) {
    auto* ep = (Endpoint*)h;
    std::unique_lock<std::mutex> g(ep->mu);
    // wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer intercepts; wait_for uses
    // pthread_cond_clockwait, which gcc 11's libtsan does not (it then believes the mutex stays held across the wait
    // and reports a "double lock" plus a false race on every queue access)
    if (!ep->cv.wait_until(g, std::chrono::system_clock::now() + std::chrono::milliseconds(timeout_ms),
                           [&] { return !ep->queue.empty(); }))
        return EV_TIMEOUT;
    ep->current = std::move(ep->queue.front());
    ep->queue.pop_front();
    *conn = ep->current.conn;
    *tag = ep->current.tag;
    *size = ep->current.payload.size();
    return ep->current.kind;
}

EP_API int ddl_ep_take(void* h, void* buf, uint64_t cap) {
    auto* ep = (Endpoint*)h;
    if (cap < ep->current.payload.size()) { g_err = "buffer too small"; return -1; }
    memcpy(buf, ep->current.payload.data(), ep->current.payload.size());
    ep->current.payload = std::vector<char>();
    return 0;
}

EP_API int ddl_ep_num_conns(void* h) {
    auto* ep = (Endpoint*)h;
    std::lock_guard<std::mutex> g(ep->mu);
    return (int)ep->conns.size();
}

EP_API void ddl_ep_close_conn(void* h, int conn) { drop_conn((Endpoint*)h, conn, false); }

EP_API void ddl_ep_destroy(void* h) {
    auto* ep = (Endpoint*)h;
    ep->stop.store(true);
    uint64_t one = 1;
    (void)!write(ep->wakefd, &one, sizeof(one));
    if (ep->loop.joinable()) ep->loop.join();
    std::vector<int> ids;
    {
        std::lock_guard<std::mutex> g(ep->mu);
        for (auto& kv : ep->conns) ids.push_back(kv.first);
    }
    for (int id : ids) drop_conn(ep, id, false);
    if (ep->listenfd >= 0) {
        close(ep->listenfd);
        unlink(ep->listen_path.c_str());
    }
    close(ep->wakefd);
    close(ep->epfd);
    delete ep;
}
