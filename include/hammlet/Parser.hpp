// Command-line parser with the rules of the reference's Parser (reference src/Parser.hpp:83-305):
// flags are arbitrary registered strings, several spellings form one group, a group may have default
// tokens, the first token must be a flag, a repeated flag is an error, everything that is not a
// registered flag is data for the flag before it, and a flag's tokens can be parsed again by a sub-parser.
#ifndef HAMMLET_PARSER_HPP
#define HAMMLET_PARSER_HPP

#include <initializer_list>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace hammlet {

template <typename T>
T convertType(const std::string& s) {
    T x;
    if (std::istringstream(s) >> x) return x;
    throw std::runtime_error("Conversion failed for string \"" + s + "\"!");
}

inline std::vector<std::string> tokenize(const std::string& text) {
    std::vector<std::string> out;
    std::istringstream ss(text);
    std::string t;
    while (ss >> t) out.push_back(t);
    return out;
}

class Parser {
    std::vector<std::string> mInput;
    std::map<std::string, size_t> mGroupOf;
    std::vector<std::vector<std::string>> mTokens;
    std::vector<bool> mGiven;
    std::set<std::string> mBlocked;
    bool mParsed = false;

    size_t group(const std::string& flag) const {
        auto it = mGroupOf.find(flag);
        if (it == mGroupOf.end()) throw std::runtime_error(flag + " is not registered as a flag!");
        return it->second;
    }
    void needParsed() const {
        if (!mParsed) throw std::runtime_error("Command line has not been parsed yet!");
    }

public:
    Parser(int argc, const char* argv[]) : mInput(argv + 1, argv + argc) {}
    explicit Parser(std::vector<std::string> tokens) : mInput(std::move(tokens)) {}

    void registerFlags(std::initializer_list<std::string> flags, const std::string& defaults = "") {
        if (mParsed) throw std::runtime_error("Cannot register flags, tokens have already been parsed!");
        for (const std::string& f : flags) {
            if (mGroupOf.count(f)) throw std::runtime_error("Flag " + f + " has already been registered!");
            if (mBlocked.count(f)) throw std::runtime_error("Flag " + f + " is blocked!");
            mGroupOf[f] = mTokens.size();
        }
        mTokens.push_back(tokenize(defaults));
        mGiven.push_back(false);
    }

    void parseArgs() {
        mParsed = true;
        if (mInput.empty()) return;
        if (!mGroupOf.count(mInput[0]))
            throw std::runtime_error("First input token (" + mInput[0] +
                                     ") is not a registered flag; parser does not support positional arguments!");
        size_t cur = 0;
        for (const std::string& tok : mInput) {
            auto it = mGroupOf.find(tok);
            if (it != mGroupOf.end()) {
                if (mGiven[it->second]) throw std::runtime_error("Duplicate flag " + tok + "!");
                cur = it->second;
                mGiven[cur] = true;
                mTokens[cur].clear();
            } else {
                mTokens[cur].push_back(tok);
            }
        }
        mInput.clear();
    }

    template <class T>
    T parse(const std::string& flag, size_t index = 0) {
        needParsed();
        const auto& t = mTokens[group(flag)];
        if (index >= t.size()) throw std::runtime_error("Not enough arguments for flag " + flag + "!");
        return convertType<T>(t[index]);
    }

    template <class T>
    std::vector<T> parseVector(const std::string& flag, size_t begin = 0, size_t end = 0) {
        needParsed();
        const auto& t = mTokens[group(flag)];
        if (end == 0) end = t.size();
        if (end <= begin) throw std::runtime_error("Invalid range for flag " + flag + "!");
        if (end > t.size()) throw std::runtime_error("Not enough arguments for flag " + flag + "!");
        std::vector<T> out;
        for (size_t i = begin; i < end; ++i) out.push_back(convertType<T>(t[i]));
        return out;
    }

    bool isSet(const std::string& flag) { return mGiven[group(flag)]; }
    size_t nrTokens(const std::string& flag) { needParsed(); return mTokens[group(flag)].size(); }
    std::vector<std::string> tokens(const std::string& flag) { needParsed(); return mTokens[group(flag)]; }

    // one line per flag group: "[*] flags : tokens" (reference Parser.hpp:242-269)
    void print() {
        needParsed();
        std::vector<std::vector<std::string>> names(mTokens.size());
        for (const auto& kv : mGroupOf) names[kv.second].push_back(kv.first);
        for (size_t i = 0; i < mTokens.size(); ++i) {
            std::cout << (mGiven[i] ? "[*]" : "[ ]");
            for (const auto& n : names[i]) std::cout << " " << n;
            std::cout << " :";
            for (const auto& t : mTokens[i]) std::cout << " " << t;
            std::cout << std::endl;
        }
    }

    Parser subparser(const std::string& flag) {
        Parser p(tokens(flag));
        for (const auto& kv : mGroupOf) p.mBlocked.insert(kv.first);
        return p;
    }
};

}  // namespace hammlet
#endif
