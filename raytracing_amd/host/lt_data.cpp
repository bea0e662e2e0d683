// Scene-description loader of the host mirror: D_data (data.h:15, data.cpp:676-781).
//
// A .dat file is a list of statements
//     type   <name> <type>            names a type
//     var    <type> <name> <value>    builds a variable in memory, laid out as the C
//                                     compiler of this host would lay out the same struct
//     export <name>                   ends the file; D_data returns that variable's memory
// with  <type>  = int | float | short | byte | ptr | [N]<type> | { <type> ... } | <type name>
// and   <value> = number | @constant | [ v ... ] | <raw bytes> | { v ... } | 0 | variable
//                 name | file name (a pointer to what that file exports).
// The grammar, the token rules (data.cpp:80-81,138-207: blank, \n and \r separate tokens,
// the brackets and '@' are tokens of their own, a token with a '.' that scans as %f is a
// float, "0x.." is hexadecimal, anything %d accepts is a number, the rest are names) and the
// layout rules (data.cpp:342-661: every base item is aligned to the alignment this compiler
// gives it after a char; a struct has no tail padding) are the reference's.  The
// implementation is not: the file is read whole, types are a small tree in a vector, and a
// variable's extent is found by laying the value out from offset 0 with the same alignment
// steps the value walk takes, so an array of structs whose size is not a multiple of their
// alignment gets the room the walk really uses (the reference sizes it short, data.cpp:646-651).
// Variables are caller-owned memory and are never freed, as in the reference.
#include "lt_engine.h"
#include "lt_data.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <string>
#include <vector>

namespace {

enum Tok { T_EOF, T_NAME, T_INT, T_FLOAT, T_KW_INT, T_KW_FLOAT, T_KW_SHORT, T_KW_BYTE, T_KW_PTR, T_AT, T_LSQ, T_RSQ, T_LCURL,
           T_RCURL, T_LRAW, T_RRAW, T_TYPE, T_VAR, T_EXPORT };

enum Base { B_INT, B_FLOAT, B_SHORT, B_BYTE, B_PTR, B_ARRAY, B_STRUCT };

struct Type {
    Base base;
    int count = 0;                   // B_ARRAY: elements
    int elem = -1;                   // B_ARRAY: element type
    std::vector<int> members;        // B_STRUCT
};

struct Var {
    std::string name;
    int type;
    char *data;
};

template <typename T> struct AlignOf { char c; T v; };
template <typename T> constexpr size_t align_after_char() { return offsetof(AlignOf<T>, v); }

struct Loader {
    std::string fname;               // for messages
    std::string text;
    size_t pos = 0;
    int line = 1;
    std::string tok;                 // text of the last token
    int tok_int = 0;
    float tok_float = 0.f;
    std::vector<Type> types;         // 0..4 are the base types
    std::vector<std::pair<std::string, int>> type_names;
    std::vector<Var> vars;

    [[noreturn]] void die(const char *what)
    {
        char buf[400];
        snprintf(buf, sizeof buf, "(%s:%d) %s (at \"%s\").\n", fname.c_str(), line, what, tok.c_str());
        HW_error(buf);
        exit(1);                     // a replaced HW_error may return
    }

    static bool is_filter(char c) { return c == ' ' || c == '\n' || c == '\r'; }
    static bool is_terminator(char c) { return is_filter(c) || strchr("@[]{}<>", c) != nullptr; }

    Tok next()
    {
        while (pos < text.size() && is_filter(text[pos])) {
            if (text[pos] == '\n') line++;
            pos++;
        }
        tok.clear();
        if (pos >= text.size() || text[pos] == '\xff') return T_EOF;      // a 0xff byte ends the input (data.cpp:115,146)
        if (is_terminator(text[pos])) {
            const char c = text[pos++];
            tok.assign(1, c);
            switch (c) {
            case '@': return T_AT;
            case '[': return T_LSQ;
            case ']': return T_RSQ;
            case '{': return T_LCURL;
            case '}': return T_RCURL;
            case '<': return T_LRAW;
            default: return T_RRAW;
            }
        }
        while (pos < text.size() && !is_terminator(text[pos]) && text[pos] != '\xff') tok.push_back(text[pos++]);
        static const struct { const char *s; Tok t; } kw[] = {{"int", T_KW_INT}, {"float", T_KW_FLOAT}, {"short", T_KW_SHORT},
            {"byte", T_KW_BYTE}, {"ptr", T_KW_PTR}, {"type", T_TYPE}, {"var", T_VAR}, {"export", T_EXPORT}};
        for (const auto &k : kw) if (tok == k.s) return k.t;
        if (tok.find('.') != std::string::npos && sscanf(tok.c_str(), "%f", &tok_float) == 1) return T_FLOAT;
        unsigned hex;
        if (tok.size() > 2 && tok[0] == '0' && tok[1] == 'x' && sscanf(tok.c_str() + 2, "%x", &hex) == 1) {
            tok_int = (int)hex;
            return T_INT;
        }
        if (sscanf(tok.c_str(), "%d", &tok_int) == 1) return T_INT;
        return T_NAME;
    }

    // Names are kept cut to 31 characters and looked up with the whole token, as the reference
    // does (strncpy into 32 bytes at data.cpp:718,744; strcmp with the token at :357,:739): a name
    // of up to 31 characters behaves as there.  A longer one is never found again here; in the
    // reference its 32 stored bytes have no terminator and the compare reads past them.
    int find_var(const std::string &n) const
    {
        for (size_t i = 0; i < vars.size(); i++) if (vars[i].name == n) return (int)i;
        return -1;
    }

    // <type>; -1 at the brace that closes a struct
    int parse_type()
    {
        switch (next()) {
        case T_NAME:
            for (const auto &tn : type_names) if (tn.first == tok) return tn.second;
            die("Unknown type name");
        case T_KW_INT: return B_INT;
        case T_KW_FLOAT: return B_FLOAT;
        case T_KW_SHORT: return B_SHORT;
        case T_KW_BYTE: return B_BYTE;
        case T_KW_PTR: return B_PTR;
        case T_LSQ: {
            if (next() != T_INT) die("Number expected");
            const int n = tok_int;
            // the reference takes any count; a file is not trusted with more than this many elements
            if (n < 0 || n > (1 << 24)) die("Number expected");
            if (next() != T_RSQ) die("Closing bracket expected");
            Type t;
            t.base = B_ARRAY; t.count = n;
            const int id = (int)types.size();
            types.push_back(t);
            const int el = parse_type();
            if (el < 0) die("Type definition expected");
            types[id].elem = el;
            return id;
        }
        case T_LCURL: {
            Type t;
            t.base = B_STRUCT;
            int m;
            while ((m = parse_type()) >= 0) t.members.push_back(m);
            if (t.members.empty()) die("Empty structure");
            types.push_back(t);
            return (int)types.size() - 1;
        }
        case T_RCURL: return -1;
        default: die("Type definition expected");
        }
    }

    static size_t align_of(Base b)
    {
        switch (b) {
        case B_INT: return align_after_char<int>();
        case B_FLOAT: return align_after_char<float>();
        case B_SHORT: return align_after_char<short>();
        case B_PTR: return align_after_char<char *>();
        default: return 1;
        }
    }
    static size_t size_of(Base b)
    {
        switch (b) {
        case B_INT: return sizeof(int);
        case B_FLOAT: return sizeof(float);
        case B_SHORT: return sizeof(short);
        case B_PTR: return sizeof(char *);
        default: return 1;
        }
    }

    // end offset of a value of `type` that starts at `off` (variables start on a malloc
    // boundary, so offsets and addresses align alike)
    size_t extent(int type, size_t off) const
    {
        const Type &t = types[type];
        if (t.base == B_ARRAY) {
            for (int i = 0; i < t.count; i++) off = extent(t.elem, off);
            return off;
        }
        if (t.base == B_STRUCT) {
            for (int m : t.members) off = extent(m, off);
            return off;
        }
        const size_t a = align_of(t.base);
        off = (off + a - 1) / a * a;
        return off + size_of(t.base);
    }

    // the constant behind "@name": a variable of exactly this base type
    const char *constant(Base b)
    {
        next();
        const int i = find_var(tok);
        if (i < 0) die("Unknown name");
        if (vars[i].type != (int)b) die(b == B_PTR ? "Expected to be a constant pointer" : "Expected to be a constant");
        return vars[i].data;
    }

    // reads one value of `type` into base[off..]; returns the end offset
    size_t parse_value(int type, char *base, size_t off)
    {
        const Type &t = types[type];
        if (t.base == B_ARRAY) {
            const Tok open = next();
            if (open == T_LRAW) {                    // raw bytes: the N characters after '<', whatever they are
                if (types[t.elem].base != B_BYTE) die("Raw array is allowed only for bytes");
                for (int i = 0; i < t.count; i++) {
                    const char c = pos < text.size() ? text[pos++] : '\xff';
                    base[off++] = c;
                }
                if (next() != T_RRAW) die("Closing bracket expected");
                return off;
            }
            if (open != T_LSQ) die("Opening bracket expected");
            for (int i = 0; i < t.count; i++) off = parse_value(t.elem, base, off);
            if (next() != T_RSQ) die("Closing bracket expected");
            return off;
        }
        if (t.base == B_STRUCT) {
            if (next() != T_LCURL) die("Opening bracket expected");
            for (int m : t.members) off = parse_value(m, base, off);
            if (next() != T_RCURL) die("Closing bracket expected");
            return off;
        }
        const size_t a = align_of(t.base);
        off = (off + a - 1) / a * a;
        char *at = base + off;
        const Tok k = next();
        switch (t.base) {
        case B_BYTE:
            if (k == T_INT) *at = (char)tok_int;
            else if (k == T_AT) *at = *constant(B_BYTE);
            else die("Constant expected");
            break;
        case B_SHORT:
            if (k == T_INT) { const short v = (short)tok_int; memcpy(at, &v, sizeof v); }
            else if (k == T_AT) memcpy(at, constant(B_SHORT), sizeof(short));
            else die("Constant expected");
            break;
        case B_INT:
            if (k == T_INT) memcpy(at, &tok_int, sizeof(int));
            else if (k == T_AT) memcpy(at, constant(B_INT), sizeof(int));
            else die("Constant expected");
            break;
        case B_FLOAT:
            if (k == T_FLOAT) memcpy(at, &tok_float, sizeof(float));
            else if (k == T_INT) { const float v = (float)tok_int; memcpy(at, &v, sizeof v); }
            else if (k == T_AT) memcpy(at, constant(B_FLOAT), sizeof(float));
            else die("Constant expected");
            break;
        default: {                                   // B_PTR
            char *p = nullptr;
            if (k == T_INT) {
                if (tok_int != 0) die("Only \"0\" allowed as a pointer");
            } else if (k == T_NAME) {
                const int i = find_var(tok);
                if (i >= 0) p = vars[i].data;
                else {                               // not a variable: a file, and the pointer is what it exports
                    if (tok == fname) die("Can't recurse files");
                    p = (char *)D_data((char *)std::string(tok).c_str());
                }
            } else if (k == T_AT) {
                memcpy(&p, constant(B_PTR), sizeof p);
            } else die("Name or \"0\" expected");
            memcpy(at, &p, sizeof p);
            break;
        }
        }
        return off + size_of(t.base);
    }

    void *run(const char *name)
    {
        fname.assign(name);
        if (fname.size() > 31) fname.resize(31);     // the reference keeps 32 characters of the name (data.cpp:28,687)
        FILE *f = fopen(name, "rb");
        if (!f) {
            char buf[300];
            snprintf(buf, sizeof buf, "(Data) Can't open %s.\n", name);
            HW_error(buf);
            exit(1);
        }
        char chunk[4096];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) text.append(chunk, got);
        fclose(f);
        for (int b = B_INT; b <= B_PTR; b++) {
            Type t;
            t.base = (Base)b;
            types.push_back(t);
        }
        for (;;) {
            switch (next()) {
            case T_TYPE: {
                if (next() != T_NAME) die("Name expected");
                for (const auto &tn : type_names) if (tn.first == tok) die("Duplicate type name");
                const std::string n = tok.substr(0, 31);
                const int t = parse_type();
                if (t < 0) die("Type definition expected");
                type_names.emplace_back(n, t);
                break;
            }
            case T_VAR: {
                const int t = parse_type();
                if (t < 0) die("Type definition expected");
                if (next() != T_NAME) die("Name expected");
                if (find_var(tok) >= 0) die("Duplicate variable name");
                Var v;
                v.name = tok.substr(0, 31);
                v.type = t;
                size_t bytes = extent(t, 0);
                v.data = (char *)calloc(bytes ? bytes : 1, 1);
                if (!v.data) die("Not enough memory");
                vars.push_back(v);                   // known while its own value is read, as in data.cpp:744-747
                const int idx = (int)vars.size() - 1;
                parse_value(t, vars[idx].data, 0);
                break;
            }
            case T_EXPORT: {
                if (next() != T_NAME) die("Name expected");
                const int i = find_var(tok);
                if (i < 0) die("Unknown variable name");
                return vars[i].data;
            }
            default: die("No such statement");
            }
        }
    }
};

}  // namespace

void *D_data(char *name)
{
    Loader l;
    return l.run(name);
}
