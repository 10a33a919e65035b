// ref_config — C shim around the reference's own ini reader, compiled from the sources where they lie under
// /root/reference (never copied): src/ConfigParser/ConfigParser.cpp and Include/ConfigParser/ConfigParser.hpp.
//
// TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libref_config.so by oracle/Makefile, in the build container.
// The reference's source includes its header as "configparser.hpp" (lower case); the Makefile puts a symbolic link of
// that name to the reference's header into oracle/_ref/inc, so the header compiled is the reference's own.
//
// The parser keeps its table private, so the shim answers queries the way src/main.cpp:339-349 asks them:
// aConfigVec<std::string>(section, name), plus the typed scalar getters.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "configparser.hpp"

namespace {
struct RefIni {
    ConfigParser* parser = nullptr;
    std::string error;      // what() of the exception the constructor threw, if it did
    std::string scratch;
};
}  // namespace

extern "C" {

void* ref_ini_open(const char* path) {
    RefIni* r = new RefIni();
    try { r->parser = new ConfigParser(path); }
    catch (const std::runtime_error& e) { r->error = e.what(); }
    return r;
}
void ref_ini_close(void* h) { RefIni* r = (RefIni*)h; delete r->parser; delete r; }
const char* ref_ini_error(void* h) { return ((RefIni*)h)->error.c_str(); }

// Number of values under (section, name); 0 for a key the file does not define. The values, each terminated by
// '\n' (they never contain one: the parser removes all white space), are left for ref_ini_text.
int32_t ref_ini_values(void* h, const char* section, const char* name) {
    RefIni* r = (RefIni*)h;
    if (!r->parser) return -1;
    std::vector<std::string> v = r->parser->aConfigVec<std::string>(section, name);
    r->scratch.clear();
    for (const std::string& s : v) { r->scratch += s; r->scratch += '\n'; }
    return (int32_t)v.size();
}
const char* ref_ini_text(void* h) { return ((RefIni*)h)->scratch.c_str(); }

// Typed access at a position that exists (the reference indexes without a bounds check).
int32_t ref_ini_int(void* h, const char* section, const char* name, uint32_t pos) { return ((RefIni*)h)->parser->aConfig<int>(section, name, pos); }
float ref_ini_float(void* h, const char* section, const char* name, uint32_t pos) { return ((RefIni*)h)->parser->aConfig<float>(section, name, pos); }
int32_t ref_ini_bool(void* h, const char* section, const char* name, uint32_t pos) { return ((RefIni*)h)->parser->aConfig<bool>(section, name, pos) ? 1 : 0; }

}  // extern "C"
