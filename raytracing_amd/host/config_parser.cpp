#include "config_parser.h"

#include <cctype>
#include <fstream>
#include <stdexcept>

namespace rta {

ConfigParser::ConfigParser(const std::string& path) {
    std::ifstream in(path);
    std::string raw, section;
    size_t lineNo = 1;
    while (std::getline(in, raw)) {
        std::string line;
        for (unsigned char ch : raw) if (!std::isspace(ch)) line.push_back((char)ch);
        if (line.empty() || line[0] == ';') { ++lineNo; continue; }
        if (line.front() == '[' && line.back() == ']') { section = line.substr(1, line.size() - 2); continue; }
        size_t eq = line.find('=');
        if (eq == std::string::npos) throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": parsing error \n" + line);
        std::string key = line.substr(0, eq), rest = line.substr(eq + 1);
        // split on ',' the way the reference's getline loop does: "a,b" -> a b ; "" -> one empty value ; "a," -> a and an empty value
        std::vector<std::string> values;
        size_t pos = 0;
        for (;;) {
            size_t comma = rest.find(',', pos);
            if (comma == std::string::npos) { values.push_back(rest.substr(pos)); break; }
            values.push_back(rest.substr(pos, comma - pos));
            pos = comma + 1;
        }
        entries_.insert({section + " - " + key, values});     // insert: an earlier definition is kept
        ++lineNo;
    }
}

bool ConfigParser::Has(const std::string& section, const std::string& name) const { return entries_.count(section + " - " + name) != 0; }

std::vector<std::string> ConfigParser::Values(const std::string& section, const std::string& name) const {
    auto it = entries_.find(section + " - " + name);
    return it == entries_.end() ? std::vector<std::string>() : it->second;
}

}  // namespace rta
