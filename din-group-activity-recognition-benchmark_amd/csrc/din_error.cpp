// Error plumbing + build identification for libdin_hip.so
#include "din_common.h"
#include <string.h>
#include <map>
#include <mutex>
#include <string>

static thread_local char g_err[512] = "";

void din_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- options (din_common.h: DIN_OPT) ----------------------------------------------------------------------------------------------
// name -> slot; slots and value strings are never freed (a reader may still hold the pointer it loaded), a few hundred bytes per process.
static std::mutex g_opt_mutex;
static std::map<std::string, din_option_slot*>& opt_table() { static std::map<std::string, din_option_slot*> t; return t; }

din_option_slot* din_option_register(const char* name) {
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    din_option_slot*& s = opt_table()[name];
    if (!s) { s = new din_option_slot; s->value.store(nullptr, std::memory_order_release); }
    return s;
}

extern "C" {
int din_set_option(const char* name, const char* value) {
    DIN_REQUIRE(name && strncmp(name, "DIN_", 4) == 0 && strlen(name) < 64, "set_option: option names start with DIN_");
    DIN_REQUIRE(!value || strlen(value) < 256, "set_option: value too long");
    din_option_slot* s = din_option_register(name);
    char* copy = nullptr;
    if (value) { copy = new char[strlen(value) + 1]; strcpy(copy, value); }
    s->value.store(copy, std::memory_order_release);
    return DIN_OK;
}
int din_get_option(const char* name, char* buf, int buf_bytes) {
    DIN_REQUIRE(name && buf && buf_bytes > 0, "get_option: null pointer");
    const char* v = din_option_register(name)->value.load(std::memory_order_acquire);
    if (!v) { buf[0] = 0; return 0; }
    snprintf(buf, (size_t)buf_bytes, "%s", v);
    return 1;
}
int din_abi_version(void) { return DIN_ABI_VERSION; }
const char* din_last_error_string(void) { return g_err; }
const char* din_build_arch(void) { return "gfx950"; }
}
