// INI reader with the reference's rules (Include/ConfigParser/ConfigParser.hpp, src/ConfigParser/ConfigParser.cpp):
//   - ALL whitespace is removed from every line before it is looked at (paths cannot contain blanks)
//   - empty lines and lines starting with ';' are skipped; `[name]` opens a section
//   - `key=v1,v2,...` stores the comma-separated values under "section - key"; the first definition
//     of a key wins; any other line is an error
#pragma once

#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace rta {

class ConfigParser {
public:
    // Throws std::runtime_error("<path>:<line>: parsing error ...") on a malformed line.
    explicit ConfigParser(const std::string& path);

    bool Has(const std::string& section, const std::string& name) const;
    // Values of a key; empty when the key does not exist.
    std::vector<std::string> Values(const std::string& section, const std::string& name) const;

    // Typed access like the reference's aConfig<T> / aConfigVec<T>: stream extraction, hex when the
    // text contains "0x". A missing key or position yields a value-initialised T.
    template <typename T>
    T Get(const std::string& section, const std::string& name, size_t pos = 0) const {
        T out{};
        std::vector<std::string> v = Values(section, name);
        if (pos < v.size()) Extract(v[pos], out);
        return out;
    }
    template <typename T>
    std::vector<T> GetAll(const std::string& section, const std::string& name) const {
        std::vector<std::string> v = Values(section, name);
        std::vector<T> out(v.size());
        for (size_t i = 0; i < v.size(); ++i) Extract(v[i], out[i]);
        return out;
    }

private:
    template <typename T>
    static void Extract(const std::string& text, T& out) {
        std::istringstream iss(text);
        if (text.find("0x") != std::string::npos) iss >> std::hex >> out; else iss >> std::dec >> out;
    }
    static void Extract(const std::string& text, bool& out) { out = text == "true" || text == "TRUE" || text == "1"; }

    std::map<std::string, std::vector<std::string>> entries_;
};

}  // namespace rta
