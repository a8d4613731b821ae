// ORACLE (test infrastructure, NOT product code) -- restatement of the reference VM: assembler, program blocks and
// program hash, decoder and user stack, for the WHOLE instruction set and all four block kinds (Span / Group / Switch / Loop).
// It produces the execution traces the prover parity tests feed to the hot path; tests/test_oracle_isa.py pins it against the
// reference's own fixtures (src/tests/mod.rs, src/tests/comparisons.rs, src/processor/mod.rs:190-346, src/processor/stack/tests/*,
// src/programs/tests/mod.rs, src/programs/assembly/tests.rs).
//
// Follows: /root/reference/src/utils/sponge.rs:13-65 (accumulator round), src/utils/hasher.rs:12-90 (RESCR round),
// src/processor/opcodes.rs:5-86 (encodings), src/programs/assembly/mod.rs:19-48,133-199,253-290 (compile, parse_branch,
// add_span, repeat), src/programs/assembly/parsers.rs:44-62 (push alignment), src/programs/blocks/mod.rs:88-218
// (Span, Group hash), src/programs/hashing.rs:15-74, src/programs/mod.rs:35-55 (program hash),
// src/processor/mod.rs:23-179 (execute, execute_loop), src/processor/decoder/mod.rs (trace builder), src/processor/stack/mod.rs
// (every operation, secret tapes, execution hints), src/programs/assembly/parsers.rs (every instruction and macro),
// src/programs/blocks/mod.rs:231-367 (Switch, Loop, validate_block_list), src/utils/hasher.rs:12-26 (digest).
// Panics of the reference are std::runtime_error with the reference's message text.
#pragma once
#include "field.hpp"
#include "rescue_constants.hpp"
#include <string>
#include <sstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <algorithm>
#include <array>

namespace orc {

static inline u128 cst(const uint64_t v[2]) { return make_u128(v[1], v[0]); }

static const size_t BASE_CYCLE_LENGTH = 16;     // lib.rs:85
static const size_t SPONGE_WIDTH = 4;           // lib.rs:104
static const size_t HASH_STATE_WIDTH = 6;       // lib.rs:98
static const size_t HACC_NUM_ROUNDS = 14;       // lib.rs:106
static const size_t MIN_TRACE_LENGTH = 16;      // lib.rs:82
static const size_t MIN_STACK_DEPTH = 8, MIN_CONTEXT_DEPTH = 1, MIN_LOOP_DEPTH = 1;   // lib.rs:87-89
static const size_t MAX_CONTEXT_DEPTH = 16, MAX_LOOP_DEPTH = 8, MAX_STACK_DEPTH = 32; // lib.rs:80,81,138
static const size_t MAX_PUBLIC_INPUTS = 8, MAX_REGISTER_COUNT = 128;                  // lib.rs:136,83
static const size_t PROGRAM_DIGEST_SIZE = 2;    // lib.rs:105
static const u128 RESCUE_INV_ALPHA = make_u128(0xAAAAAAAAAAAAAAAAull, 0xAAAA8CAAAAAAAAABull);  // sponge.rs:70

// ---- Rescue building blocks (generic width) -----------------------------------------------------------
template <size_t W> struct Rescue {
    static void sbox(u128* s) { for (size_t i = 0; i < W; i++) s[i] = exp(s[i], 3); }
    static void inv_sbox(u128* s) { for (size_t i = 0; i < W; i++) s[i] = exp(s[i], RESCUE_INV_ALPHA); }
    static void matmul(u128* s, const uint64_t (*m)[2]) {
        u128 r[W];
        for (size_t i = 0; i < W; i++) {
            u128 acc = 0;
            for (size_t j = 0; j < W; j++) acc = add(acc, mul(cst(m[i * W + j]), s[j]));
            r[i] = acc;
        }
        for (size_t i = 0; i < W; i++) s[i] = r[i];
    }
};
static inline void sponge_mds(u128* s) { Rescue<4>::matmul(s, SPONGE_MDS); }           // sponge.rs:52
static inline void sponge_inv_mds(u128* s) { Rescue<4>::matmul(s, SPONGE_INV_MDS); }   // sponge.rs:66 (apply_inv_mds)
static inline void hasher_mds(u128* s) { Rescue<6>::matmul(s, HASHER_MDS); }
static inline void hasher_inv_mds(u128* s) { Rescue<6>::matmul(s, HASHER_INV_MDS); }

// modified Rescue round with op_code / op_value injection                 sponge.rs:13-30
static inline void sponge_apply_round(u128* state, u128 op_code, u128 op_value, size_t step) {
    size_t idx = step % BASE_CYCLE_LENGTH;
    for (size_t i = 0; i < 4; i++) state[i] = add(state[i], cst(SPONGE_ARK[i][idx]));
    Rescue<4>::sbox(state);
    sponge_mds(state);
    state[0] = add(state[0], op_code);
    state[1] = add(state[1], op_value);
    for (size_t i = 0; i < 4; i++) state[i] = add(state[i], cst(SPONGE_ARK[4 + i][idx]));
    Rescue<4>::inv_sbox(state);
    sponge_mds(state);
}
// plain Rescue round on the 6-element hasher state                         hasher.rs:28-40
static inline void hasher_apply_round(u128* state, size_t step) {
    size_t idx = step % BASE_CYCLE_LENGTH;
    for (size_t i = 0; i < 6; i++) state[i] = add(state[i], cst(HASHER_ARK[i][idx]));
    Rescue<6>::sbox(state);
    hasher_mds(state);
    for (size_t i = 0; i < 6; i++) state[i] = add(state[i], cst(HASHER_ARK[6 + i][idx]));
    Rescue<6>::inv_sbox(state);
    hasher_mds(state);
}

// ---- opcodes (processor/opcodes.rs:5-86) ----------------------------------------------------------------
enum FlowOp : uint8_t { F_HACC = 0, F_BEGIN = 1, F_TEND = 2, F_FEND = 3, F_LOOP = 4, F_WRAP = 5, F_BREAK = 6, F_VOID = 7 };
enum UserOp : uint8_t {
    OP_ASSERT = 0x60, OP_ASSERTEQ = 0x61, OP_EQ = 0x62, OP_DROP = 0x63, OP_DROP4 = 0x64, OP_CHOOSE = 0x65, OP_CHOOSE2 = 0x66, OP_CSWAP2 = 0x67,
    OP_ADD = 0x68, OP_MUL = 0x69, OP_AND = 0x6A, OP_OR = 0x6B, OP_INV = 0x6C, OP_NEG = 0x6D, OP_NOT = 0x6E,
    OP_READ = 0x70, OP_READ2 = 0x71, OP_DUP = 0x72, OP_DUP2 = 0x73, OP_DUP4 = 0x74, OP_PAD2 = 0x75,
    OP_SWAP = 0x78, OP_SWAP2 = 0x79, OP_SWAP4 = 0x7A, OP_ROLL4 = 0x7B, OP_ROLL8 = 0x7C, OP_BINACC = 0x7D,
    OP_PUSH = 0x1F, OP_CMP = 0x3F, OP_RESCR = 0x5F, OP_BEGIN = 0x00, OP_NOOP = 0x7F,
};
static inline size_t ld_index(UserOp op) { return (size_t)op & 0x1F; }          // opcodes.rs:90
static inline size_t hd_index(UserOp op) { return ((size_t)op >> 5) & 3; }      // opcodes.rs:101

static inline const char* op_name(UserOp op) {                                    // opcodes.rs:117-164 (Display)
    switch (op) {
        case OP_BEGIN: return "begin"; case OP_NOOP: return "noop"; case OP_ASSERT: return "assert"; case OP_ASSERTEQ: return "asserteq";
        case OP_PUSH: return "push"; case OP_READ: return "read"; case OP_READ2: return "read2"; case OP_DUP: return "dup"; case OP_DUP2: return "dup2";
        case OP_DUP4: return "dup4"; case OP_PAD2: return "pad2"; case OP_DROP: return "drop"; case OP_DROP4: return "drop4"; case OP_SWAP: return "swap";
        case OP_SWAP2: return "swap2"; case OP_SWAP4: return "swap4"; case OP_ROLL4: return "roll4"; case OP_ROLL8: return "roll8";
        case OP_CHOOSE: return "choose"; case OP_CHOOSE2: return "choose2"; case OP_CSWAP2: return "cswap2"; case OP_ADD: return "add";
        case OP_MUL: return "mul"; case OP_INV: return "inv"; case OP_NEG: return "neg"; case OP_NOT: return "not"; case OP_AND: return "and";
        case OP_OR: return "or"; case OP_EQ: return "eq"; case OP_CMP: return "cmp"; case OP_BINACC: return "binacc"; case OP_RESCR: return "rescr";
    }
    return "?";
}
static inline bool is_user_op(uint8_t code) {
    switch ((UserOp)code) {
        case OP_BEGIN: case OP_NOOP: case OP_ASSERT: case OP_ASSERTEQ: case OP_PUSH: case OP_READ: case OP_READ2: case OP_DUP: case OP_DUP2: case OP_DUP4:
        case OP_PAD2: case OP_DROP: case OP_DROP4: case OP_SWAP: case OP_SWAP2: case OP_SWAP4: case OP_ROLL4: case OP_ROLL8: case OP_CHOOSE: case OP_CHOOSE2:
        case OP_CSWAP2: case OP_ADD: case OP_MUL: case OP_INV: case OP_NEG: case OP_NOT: case OP_AND: case OP_OR: case OP_EQ: case OP_CMP: case OP_BINACC:
        case OP_RESCR: return true;
    }
    return false;
}

// execution hints (processor/opcodes.rs:168-201)
enum HintKind : uint8_t { H_NONE = 0, H_EQ_START = 1, H_RC_START = 2, H_CMP_START = 3, H_PMPATH_START = 4, H_PUSH_VALUE = 5 };
struct OpHint {
    HintKind kind = H_NONE; u128 v = 0;
    u128 value() const { return kind == H_PUSH_VALUE ? v : 0; }                 // opcodes.rs:180
    static OpHint push(u128 x) { OpHint h; h.kind = H_PUSH_VALUE; h.v = x; return h; }
    static OpHint of(HintKind k, u128 x = 0) { OpHint h; h.kind = k; h.v = x; return h; }
};
static inline std::string u128_to_string(u128 v) {
    if (v == 0) return "0";
    std::string s;
    while (v) { s.insert(s.begin(), char('0' + (int)(v % 10))); v /= 10; }
    return s;
}
static inline std::string hint_display(const OpHint& h) {                          // opcodes.rs:188-199
    switch (h.kind) {
        case H_EQ_START: return "::eq";
        case H_RC_START: case H_CMP_START: case H_PMPATH_START: return "." + u128_to_string(h.v);
        case H_PUSH_VALUE: return "(" + u128_to_string(h.v) + ")";
        default: return "";
    }
}

// ---- program blocks (programs/blocks/mod.rs) -------------------------------------------------------------------
typedef std::map<size_t, OpHint> HintMap;
struct Span {
    std::vector<UserOp> ops; HintMap hints;
    Span() {}
    Span(const std::vector<UserOp>& instructions, const HintMap& h) : ops(instructions), hints(h) {      // blocks/mod.rs:90
        if (ops.size() % BASE_CYCLE_LENGTH != BASE_CYCLE_LENGTH - 1)
            throw std::runtime_error("invalid number of instructions: expected one less than a multiple of 16, but was " + std::to_string(ops.size()));
        for (size_t i = 0; i < ops.size(); i++) {
            if (ops[i] != OP_PUSH) continue;
            if (i % 8 != 0) throw std::runtime_error("PUSH is not allowed on step " + std::to_string(i) + ", must be on step which is a multiple of 8");
            auto it = hints.find(i);
            if (it == hints.end()) throw std::runtime_error("invalid PUSH operation on step " + std::to_string(i) + ": operation value is missing");
            if (it->second.kind != H_PUSH_VALUE) throw std::runtime_error("invalid PUSH operation on step " + std::to_string(i) + ": operation value is of wrong type");
        }
        for (auto& kv : hints) if (kv.first >= ops.size()) throw std::runtime_error("hint out of bounds");
    }
    size_t length() const { return ops.size(); }
    OpHint get_hint(size_t i) const { auto it = hints.find(i); return it == hints.end() ? OpHint() : it->second; }   // blocks/mod.rs:142
    bool starts_with(const std::vector<UserOp>& p) const { return ops.size() >= p.size() && std::equal(p.begin(), p.end(), ops.begin()); }
    std::string debug() const {                                                  // blocks/mod.rs:181
        std::string s;
        for (size_t i = 0; i < ops.size(); i++) { if (i) s += " "; s += op_name(ops[i]); s += hint_display(get_hint(i)); }
        return s;
    }
};
enum BlockKind : uint8_t { B_SPAN, B_GROUP, B_SWITCH, B_LOOP };
struct Block {
    BlockKind kind = B_SPAN;
    Span span;
    std::vector<Block> body;     // Group body | Switch true branch | Loop body
    std::vector<Block> alt;      // Switch false branch | Loop skip block
    bool is_span() const { return kind == B_SPAN; }
};

static inline void hash_op(u128* state, uint8_t op_code, u128 op_value, size_t step) { sponge_apply_round(state, op_code, op_value, step); }  // hashing.rs:62
static inline std::array<u128, 4> hash_acc(u128 parent_hash, u128 v0, u128 v1) {      // hashing.rs:68
    std::array<u128, 4> st = {parent_hash, v0, v1, 0};
    for (size_t i = 1; i < 1 + HACC_NUM_ROUNDS; i++) hash_op(st.data(), OP_NOOP, 0, i);
    return st;
}
static inline void span_hash(const Span& s, u128* state) {                          // blocks/mod.rs:149
    for (size_t i = 0; i < s.ops.size(); i++) {
        u128 v = 0;
        if (s.ops[i] == OP_PUSH) {
            OpHint h = s.get_hint(i);
            if (h.kind != H_PUSH_VALUE) throw std::runtime_error("value for PUSH operation is missing");
            v = h.v;
        }
        hash_op(state, s.ops[i], v, i);
    }
}
static const std::vector<uint8_t> BLOCK_SUFFIX = {OP_NOOP};                          // blocks/mod.rs:9
static const size_t BLOCK_SUFFIX_OFFSET = BASE_CYCLE_LENGTH - 1;                     // blocks/mod.rs:10
static inline std::vector<UserOp> loop_skip_ops() { std::vector<UserOp> v(15, OP_NOOP); v[0] = OP_NOT; v[1] = OP_ASSERT; return v; }   // blocks/mod.rs:12
static inline std::vector<uint8_t> loop_block_suffix() { std::vector<uint8_t> v(16, OP_NOOP); v[0] = OP_NOT; v[1] = OP_ASSERT; return v; }   // blocks/mod.rs:19

static std::pair<u128, u128> block_hash(const Block& b);
static u128 hash_seq(const std::vector<Block>& blocks, const std::vector<uint8_t>& suffix, size_t suffix_offset) {   // hashing.rs:15
    u128 st[4] = {0, 0, 0, 0};
    if (!blocks[0].is_span()) throw std::runtime_error("first block in a sequence must be a Span block");
    span_hash(blocks[0].span, st);
    for (size_t b = 1; b < blocks.size(); b++) {
        if (blocks[b].is_span()) {
            hash_op(st, OP_NOOP, 0, BASE_CYCLE_LENGTH - 1);
            span_hash(blocks[b].span, st);
        } else {
            auto h = block_hash(blocks[b]);
            auto m = hash_acc(st[0], h.first, h.second);
            for (int i = 0; i < 4; i++) st[i] = m[i];
        }
    }
    for (size_t i = 0; i < suffix.size(); i++) hash_op(st, suffix[i], 0, suffix_offset + i);
    return st[0];
}
static inline u128 seq_hash(const std::vector<Block>& blocks) { return hash_seq(blocks, BLOCK_SUFFIX, BLOCK_SUFFIX_OFFSET); }
static inline u128 loop_image(const Block& l) { return hash_seq(l.body, {}, 0); }                 // blocks/mod.rs:306
static inline u128 loop_body_hash(const Block& l) { return hash_seq(l.body, loop_block_suffix(), 0); }   // blocks/mod.rs:310
static inline u128 loop_skip_hash(const Block& l) { return seq_hash(l.alt); }                     // blocks/mod.rs:318
static std::pair<u128, u128> block_hash(const Block& b) {
    switch (b.kind) {
        case B_GROUP: return {seq_hash(b.body), 0};                                // blocks/mod.rs:215
        case B_SWITCH: return {seq_hash(b.body), seq_hash(b.alt)};                 // blocks/mod.rs:264
        case B_LOOP: return {loop_body_hash(b), loop_skip_hash(b)};                // blocks/mod.rs:322
        default: return {0, 0};
    }
}
static inline void validate_block_list(const std::vector<Block>& blocks, const std::vector<UserOp>& starts_with) {   // blocks/mod.rs:341
    if (blocks.empty()) throw std::runtime_error("a sequence of blocks must contain at least one block");
    if (!blocks[0].is_span()) throw std::runtime_error("a sequence of blocks must start with a Span block");
    if (!starts_with.empty() && !blocks[0].span.starts_with(starts_with)) throw std::runtime_error("the first block does not start with a valid sequence of instructions");
    bool was_span = true;
    for (size_t i = 1; i < blocks.size(); i++) {
        if (blocks[i].is_span()) { if (was_span) throw std::runtime_error("a Span block cannot be followed by another Span block"); }
        else was_span = false;
    }
}
static inline Block make_span(const std::vector<UserOp>& ops, const HintMap& hints = HintMap()) { Block b; b.kind = B_SPAN; b.span = Span(ops, hints); return b; }
static inline Block make_group(const std::vector<Block>& body) { validate_block_list(body, {}); Block b; b.kind = B_GROUP; b.body = body; return b; }   // blocks/mod.rs:198
static inline Block make_switch(const std::vector<Block>& t, const std::vector<Block>& f) {          // blocks/mod.rs:235
    validate_block_list(t, {OP_ASSERT}); validate_block_list(f, {OP_NOT, OP_ASSERT});
    Block b; b.kind = B_SWITCH; b.body = t; b.alt = f; return b;
}
static inline Block make_loop(const std::vector<Block>& body) {                                      // blocks/mod.rs:289
    validate_block_list(body, {OP_ASSERT});
    Block b; b.kind = B_LOOP; b.body = body; b.alt = {make_span(loop_skip_ops())}; return b;
}
static std::string block_debug(const Block& b) {                                                     // blocks/mod.rs:74,221,271,329
    auto seq = [](const std::vector<Block>& v) { std::string s; for (auto& x : v) s += block_debug(x) + " "; return s; };
    switch (b.kind) {
        case B_SPAN: return b.span.debug();
        case B_GROUP: return "block " + seq(b.body) + "end";
        case B_SWITCH: return "if " + seq(b.body) + "else " + seq(b.alt) + "end";
        case B_LOOP: return "while " + seq(b.body) + "end";
    }
    return "";
}

struct Program {
    Block root;                  // Group
    u128 hash[2];
    void finalize() {            // programs/mod.rs:35-55
        if (root.body.empty() || !root.body[0].is_span()) throw std::runtime_error("a program must start with a Span block");
        if (root.body[0].span.ops[0] != OP_BEGIN) throw std::runtime_error("a program must start with BEGIN operation");
        auto h = block_hash(root);
        auto acc = hash_acc(0, h.first, h.second);
        hash[0] = acc[0]; hash[1] = acc[1];
    }
    static Program from_root_blocks(const std::vector<Block>& blocks) { Program p; p.root = make_group(blocks); p.finalize(); return p; }
    std::string debug() const { std::string s = block_debug(root); return s.substr(6); }             // programs/mod.rs:68 (drops "block ")
};

// ---- assembler (programs/assembly/mod.rs, parsers.rs) ------------------------------------------------------------------
struct Assembler {
    std::vector<std::string> tokens;
    static const size_t PUSH_OP_ALIGNMENT = 8, HASH_OP_ALIGNMENT = 16;            // parsers.rs:6-7

    static void add_span(std::vector<Block>& body, std::vector<UserOp>& ops, HintMap& hints, bool force) {  // assembly/mod.rs:253
        if (ops.empty() && !force) return;
        std::vector<UserOp> span_ops = ops;
        size_t pad = BASE_CYCLE_LENGTH - (span_ops.size() % BASE_CYCLE_LENGTH) - 1;
        span_ops.resize(span_ops.size() + pad, OP_NOOP);
        body.push_back(make_span(span_ops, hints));
        ops.clear(); hints.clear();
    }
    static Block merge_spans(const Block& a, const Block& b) {             // blocks/mod.rs:163
        if (!a.is_span() || !b.is_span()) throw std::runtime_error("merge_spans: not a Span block");
        std::vector<UserOp> ops = a.span.ops;
        ops.push_back(OP_NOOP);
        ops.insert(ops.end(), b.span.ops.begin(), b.span.ops.end());
        HintMap hints = a.span.hints;
        size_t off = a.span.ops.size() + 1;
        for (auto& kv : b.span.hints) hints[kv.first + off] = kv.second;
        return make_span(ops, hints);
    }
    static std::vector<Block> repeat_block_sequence(const std::vector<Block>& tpl, size_t n) {   // assembly/mod.rs:271
        std::vector<Block> body;
        if (!tpl.back().is_span()) { for (size_t i = 0; i < n; i++) body.insert(body.end(), tpl.begin(), tpl.end()); }
        else {
            body = tpl;
            for (size_t i = 1; i < n; i++) {
                body.back() = merge_spans(body.back(), tpl[0]);
                body.insert(body.end(), tpl.begin() + 1, tpl.end());
            }
        }
        return body;
    }
    static std::vector<std::string> split_dot(const std::string& s) {
        std::vector<std::string> r; size_t p = 0;
        for (;;) { size_t q = s.find('.', p); if (q == std::string::npos) { r.push_back(s.substr(p)); break; } r.push_back(s.substr(p, q - p)); p = q + 1; }
        return r;
    }
    static std::string join(const std::vector<std::string>& op) { std::string s; for (size_t i = 0; i < op.size(); i++) { if (i) s += "."; s += op[i]; } return s; }
    [[noreturn]] static void err(const std::vector<std::string>& op, size_t step, const std::string& why) {
        throw std::runtime_error("assembly error at " + std::to_string(step) + ": " + join(op) + ": " + why);
    }
    static uint32_t parse_u32(const std::vector<std::string>& op, size_t step) {
        const std::string& s = op[1];
        if (s.empty() || s.size() > 10) err(op, step, "invalid parameter");
        uint64_t v = 0;
        for (char c : s) { if (c < '0' || c > '9') err(op, step, "invalid parameter"); v = v * 10 + (c - '0'); }
        if (v > 0xFFFFFFFFull) err(op, step, "invalid parameter");
        return (uint32_t)v;
    }
    static uint32_t read_param(const std::vector<std::string>& op, size_t step) {        // parsers.rs:543
        if (op.size() == 1) return 1;
        if (op.size() > 2) err(op, step, "too many parameters");
        uint32_t r = parse_u32(op, step);
        if (r == 0) err(op, step, "parameter value must be greater than 0");
        return r;
    }
    static u128 read_value(const std::vector<std::string>& op, size_t step) {            // parsers.rs:566
        if (op.size() == 1) err(op, step, "missing parameter");
        if (op.size() > 2) err(op, step, "too many parameters");
        const std::string& s = op[1];
        u128 v = 0;
        // values are parsed into 256-bit-safe form: reject anything that does not fit u128 like from_str_radix does
        if (s.rfind("0x", 0) == 0) {
            if (s.size() == 2 || s.size() > 34) err(op, step, "invalid parameter");
            for (size_t i = 2; i < s.size(); i++) {
                char c = s[i]; int d;
                if (c >= '0' && c <= '9') d = c - '0'; else if ((c | 32) >= 'a' && (c | 32) <= 'f') d = (c | 32) - 'a' + 10; else err(op, step, "invalid parameter");
                v = v * 16 + d;
            }
        } else {
            if (s.empty()) err(op, step, "invalid parameter");
            for (char c : s) {
                if (c < '0' || c > '9') err(op, step, "invalid parameter");
                u128 nv = v * 10 + (c - '0');
                if (v > (~(u128)0 - (c - '0')) / 10) err(op, step, "invalid parameter");
                v = nv;
            }
        }
        if (v >= P) err(op, step, "parameter value must be smaller than the field modulus");
        return v;
    }
    static void no_param(const std::vector<std::string>& op, size_t step) { if (op.size() > 1) err(op, step, "too many parameters"); }
    static void append_push_op(std::vector<UserOp>& p, HintMap& hints, u128 value) {       // parsers.rs:51
        size_t pad = (PUSH_OP_ALIGNMENT - p.size() % PUSH_OP_ALIGNMENT) % PUSH_OP_ALIGNMENT;
        p.resize(p.size() + pad, OP_NOOP);
        hints[p.size()] = OpHint::push(value);
        p.push_back(OP_PUSH);
    }
    static void hash_align(std::vector<UserOp>& p) {                                       // parsers.rs:426
        size_t pad = (HASH_OP_ALIGNMENT - p.size() % HASH_OP_ALIGNMENT) % HASH_OP_ALIGNMENT;
        p.resize(p.size() + pad, OP_NOOP);
    }
    static void ext(std::vector<UserOp>& p, std::initializer_list<UserOp> l) { p.insert(p.end(), l.begin(), l.end()); }
    static uint32_t bits_param(const std::vector<std::string>& op, size_t step) {
        uint32_t n = read_param(op, step);
        if (n < 4 || n > 128) err(op, step, "value must be between 4 and 128");
        return n;
    }
    static u128 pow2(uint32_t e) { return (u128)1 << e; }

    void parse_op_token(const std::vector<std::string>& op, std::vector<UserOp>& p, HintMap& hints, size_t step) {   // assembly/mod.rs:202, parsers.rs
        const std::string& name = op[0];
        if (name == "noop") { no_param(op, step); p.push_back(OP_NOOP); }                  // parsers.rs:13
        else if (name == "assert") {                                                       // :22
            if (op.size() > 2) err(op, step, "too many parameters");
            if (op.size() == 1) p.push_back(OP_ASSERT); else if (op[1] == "eq") p.push_back(OP_ASSERTEQ); else err(op, step, "allowed values are: [eq]");
        }
        else if (name == "push") append_push_op(p, hints, read_value(op, step));           // :44
        else if (name == "read") {                                                         // :65
            if (op.size() > 2) err(op, step, "too many parameters");
            if (op.size() == 1 || op[1] == "a") p.push_back(OP_READ); else if (op[1] == "ab") p.push_back(OP_READ2); else err(op, step, "allowed values are: [a, ab]");
        }
        else if (name == "dup") {                                                          // :87
            switch (read_param(op, step)) {
                case 1: p.push_back(OP_DUP); break; case 2: p.push_back(OP_DUP2); break;
                case 3: ext(p, {OP_DUP4, OP_ROLL4, OP_DROP}); break; case 4: p.push_back(OP_DUP4); break;
                default: err(op, step, "allowed values are: [1, 2, 3, 4]");
            }
        }
        else if (name == "pad") {                                                          // :102
            switch (read_param(op, step)) {
                case 1: ext(p, {OP_PAD2, OP_DROP}); break; case 2: p.push_back(OP_PAD2); break;
                case 3: ext(p, {OP_PAD2, OP_PAD2, OP_DROP}); break; case 4: ext(p, {OP_PAD2, OP_PAD2}); break;
                case 5: ext(p, {OP_PAD2, OP_PAD2, OP_PAD2, OP_DROP}); break; case 6: ext(p, {OP_PAD2, OP_PAD2, OP_PAD2}); break;
                case 7: ext(p, {OP_PAD2, OP_PAD2, OP_DUP4, OP_DROP}); break; case 8: ext(p, {OP_PAD2, OP_PAD2, OP_DUP4}); break;
                default: err(op, step, "allowed values are: [1, 2, 3, 4, 5, 6, 7, 8]");
            }
        }
        else if (name == "pick") {                                                         // :121
            switch (read_param(op, step)) {
                case 1: ext(p, {OP_DUP2, OP_DROP}); break;
                case 2: ext(p, {OP_DUP4, OP_ROLL4, OP_DROP, OP_DROP, OP_DROP}); break;
                case 3: ext(p, {OP_DUP4, OP_DROP, OP_DROP, OP_DROP}); break;
                default: err(op, step, "allowed values are: [1, 2, 3]");
            }
        }
        else if (name == "drop") {                                                         // :137
            switch (read_param(op, step)) {
                case 1: p.push_back(OP_DROP); break; case 2: ext(p, {OP_DROP, OP_DROP}); break;
                case 3: ext(p, {OP_DUP, OP_DROP4}); break; case 4: p.push_back(OP_DROP4); break;
                case 5: ext(p, {OP_DROP, OP_DROP4}); break; case 6: ext(p, {OP_DROP, OP_DROP, OP_DROP4}); break;
                case 7: ext(p, {OP_DUP, OP_DROP4, OP_DROP4}); break; case 8: ext(p, {OP_DROP4, OP_DROP4}); break;
                default: err(op, step, "allowed values are: [1, 2, 3, 4, 5, 6, 7, 8]");
            }
        }
        else if (name == "swap") {                                                         // :157
            switch (read_param(op, step)) {
                case 1: p.push_back(OP_SWAP); break; case 2: p.push_back(OP_SWAP2); break; case 4: p.push_back(OP_SWAP4); break;
                default: err(op, step, "allowed values are: [1, 2, 4]");
            }
        }
        else if (name == "roll") {                                                         // :171
            switch (read_param(op, step)) {
                case 4: p.push_back(OP_ROLL4); break; case 8: p.push_back(OP_ROLL8); break;
                default: err(op, step, "allowed values are: [4, 8]");
            }
        }
        else if (name == "add") { no_param(op, step); p.push_back(OP_ADD); }               // :187
        else if (name == "sub") { no_param(op, step); ext(p, {OP_NEG, OP_ADD}); }          // :194
        else if (name == "mul") { no_param(op, step); p.push_back(OP_MUL); }               // :201
        else if (name == "div") { no_param(op, step); ext(p, {OP_INV, OP_MUL}); }          // :208
        else if (name == "neg") { no_param(op, step); p.push_back(OP_NEG); }
        else if (name == "inv") { no_param(op, step); p.push_back(OP_INV); }
        else if (name == "not") { no_param(op, step); p.push_back(OP_NOT); }
        else if (name == "and") { no_param(op, step); p.push_back(OP_AND); }
        else if (name == "or")  { no_param(op, step); p.push_back(OP_OR); }
        else if (name == "eq") { no_param(op, step); hints[p.size()] = OpHint::of(H_EQ_START); ext(p, {OP_READ, OP_EQ}); }          // :254
        else if (name == "ne") { no_param(op, step); hints[p.size()] = OpHint::of(H_EQ_START); ext(p, {OP_READ, OP_EQ, OP_NOT}); }  // :263
        else if (name == "gt" || name == "lt") {                                           // :272, :304
            uint32_t n = bits_param(op, step);
            ext(p, {OP_PAD2, OP_PAD2, OP_PAD2, OP_DUP});
            append_push_op(p, hints, pow2(n - 1));
            hints[p.size()] = OpHint::of(H_CMP_START, n);
            p.resize(p.size() + n, OP_CMP);
            if (name == "gt") ext(p, {OP_DROP4, OP_PAD2, OP_SWAP4, OP_ROLL4, OP_ASSERTEQ, OP_ASSERTEQ, OP_ROLL4, OP_DUP, OP_DROP4});
            else ext(p, {OP_DROP4, OP_PAD2, OP_SWAP4, OP_ROLL4, OP_ASSERTEQ, OP_ASSERTEQ, OP_DUP, OP_DROP4});
        }
        else if (name == "rc") {                                                           // :335
            uint32_t n = bits_param(op, step);
            p.push_back(OP_PAD2);
            append_push_op(p, hints, 1);
            ext(p, {OP_SWAP, OP_DUP});
            hints[p.size()] = OpHint::of(H_RC_START, n);
            p.resize(p.size() + n, OP_BINACC);
            ext(p, {OP_DUP, OP_DROP4});
            hints[p.size()] = OpHint::of(H_EQ_START);
            ext(p, {OP_READ, OP_EQ});
        }
        else if (name == "isodd") {                                                        // :363
            uint32_t n = bits_param(op, step);
            p.push_back(OP_PAD2);
            append_push_op(p, hints, 1);
            ext(p, {OP_SWAP, OP_DUP});
            hints[p.size()] = OpHint::of(H_RC_START, n);
            ext(p, {OP_BINACC, OP_SWAP2, OP_ROLL4, OP_DUP});
            p.resize(p.size() + (n - 1), OP_BINACC);
            ext(p, {OP_DROP, OP_DROP, OP_SWAP, OP_ROLL4, OP_ASSERTEQ, OP_DROP});
        }
        else if (name == "choose") {                                                       // :399
            switch (read_param(op, step)) {
                case 1: p.push_back(OP_CHOOSE); break; case 2: p.push_back(OP_CHOOSE2); break;
                default: err(op, step, "allowed values are: [1, 2]");
            }
        }
        else if (name == "hash") {                                                         // :414
            switch (read_param(op, step)) {
                case 1: ext(p, {OP_PAD2, OP_PAD2, OP_PAD2, OP_DROP}); break; case 2: ext(p, {OP_PAD2, OP_PAD2}); break;
                case 3: ext(p, {OP_PAD2, OP_PAD2, OP_DROP}); break; case 4: p.push_back(OP_PAD2); break;
                default: err(op, step, "allowed values are: [1, 2, 3, 4]");
            }
            hash_align(p);
            p.resize(p.size() + 10, OP_RESCR);
            p.push_back(OP_DROP4);
        }
        else if (name == "smpath") {                                                       // :444
            uint32_t n = read_param(op, step);
            if (n < 2 || n > 256) err(op, step, "value must be between 2 and 256");
            ext(p, {OP_READ2, OP_SWAP2, OP_READ2, OP_CSWAP2, OP_PAD2});
            hash_align(p);
            const UserOp SUB[16] = {OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_RESCR, OP_DROP4, OP_READ2,
                                    OP_SWAP2, OP_READ2, OP_CSWAP2, OP_PAD2};
            for (uint32_t i = 0; i < n - 2; i++) p.insert(p.end(), SUB, SUB + 16);
            p.insert(p.end(), SUB, SUB + 11);
        }
        else if (name == "pmpath") {                                                       // :488
            uint32_t n = read_param(op, step);
            if (n < 2 || n > 256) err(op, step, "value must be between 2 and 256");
            hints[p.size()] = OpHint::of(H_PMPATH_START, n);
            ext(p, {OP_READ2, OP_PAD2});
            append_push_op(p, hints, 1);
            ext(p, {OP_SWAP, OP_DUP, OP_BINACC, OP_SWAP4, OP_CSWAP2, OP_PAD2});
            hash_align(p);
            UserOp SUB[32];
            for (int i = 0; i < 32; i++) SUB[i] = OP_NOOP;
            for (int i = 0; i < 10; i++) SUB[i] = OP_RESCR;
            const UserOp mid[9] = {OP_DROP4, OP_PAD2, OP_SWAP2, OP_READ2, OP_SWAP4, OP_BINACC, OP_SWAP4, OP_CSWAP2, OP_PAD2};
            for (int i = 0; i < 9; i++) SUB[10 + i] = mid[i];
            for (uint32_t i = 0; i < n - 2; i++) p.insert(p.end(), SUB, SUB + 32);
            p.insert(p.end(), SUB, SUB + 11);
            ext(p, {OP_SWAP2, OP_DROP, OP_ROLL4, OP_ASSERTEQ});
        }
        else err(op, step, "invalid operation");
    }
    size_t parse_block(std::vector<Block>& parent, size_t i) {            // assembly/mod.rs:54
        auto head = split_dot(tokens[i]);
        if (head[0] == "block") {
            if (head.size() > 1) err(head, i, "invalid block head");
            std::vector<Block> body;
            i = parse_branch(body, i);
            parent.push_back(make_group(body));
            return i + 1;
        }
        if (head[0] == "if") {
            if (head.size() == 1 || head[1] != "true") err(head, i, "invalid block head");
            std::vector<Block> t_branch, f_branch;
            i = parse_branch(t_branch, i);
            if (tokens[i] == "else") i = parse_branch(f_branch, i);
            else f_branch.push_back(make_span(loop_skip_ops()));              // NOT ASSERT + 13 NOOPs (assembly/mod.rs:89-94)
            parent.push_back(make_switch(t_branch, f_branch));
            return i + 1;
        }
        if (head[0] == "repeat") {
            if (head.size() != 2) err(head, i, "invalid block head");
            size_t n = parse_u32(head, i);                                  // assembly/mod.rs:306 (no default, zero allowed then rejected)
            if (n < 2) err(head, i, "invalid number of iterations");
            std::vector<Block> tpl;
            i = parse_branch(tpl, i);
            parent.push_back(make_group(repeat_block_sequence(tpl, n)));
            return i + 1;
        }
        if (head[0] == "while") {
            if (head.size() == 1 || head[1] != "true") err(head, i, "invalid block head");
            std::vector<Block> body;
            i = parse_branch(body, i);
            parent.push_back(make_loop(body));
            return i + 1;
        }
        err(head, i, "invalid block head");
    }
    size_t parse_branch(std::vector<Block>& body, size_t i) {             // assembly/mod.rs:136
        auto head = split_dot(tokens[i]);
        std::vector<UserOp> ops; HintMap hints;
        if (head[0] == "begin") { head[0] = "block"; ops = {OP_BEGIN}; }
        else if (head[0] == "block" || head[0] == "repeat") {}
        else if (head[0] == "if" || head[0] == "while") ops = {OP_ASSERT};
        else if (head[0] == "else") ops = {OP_NOT, OP_ASSERT};
        else err(head, i, "invalid block head");
        size_t first = i; i += 1;
        while (i < tokens.size()) {
            auto op = split_dot(tokens[i]);
            if (op[0] == "block" || op[0] == "if" || op[0] == "repeat" || op[0] == "while") {
                bool force_span = body.empty();
                add_span(body, ops, hints, force_span);
                i = parse_block(body, i);
            } else if (op[0] == "else") {
                if (head[0] != "if") throw std::runtime_error("dangling else at " + std::to_string(i));
                if (i - first < 2) err(head, first, "empty block");
                add_span(body, ops, hints, false);
                return i;
            } else if (op[0] == "end") {
                if (i - first < 2) err(head, first, "empty block");
                add_span(body, ops, hints, false);
                return i;
            } else { parse_op_token(op, ops, hints, i); i += 1; }
        }
        err(head, first, "unmatched block");
    }
    Program compile(const std::string& src) {                             // assembly/mod.rs:19
        std::stringstream ss(src); std::string t; tokens.clear();
        while (ss >> t) tokens.push_back(t);
        if (tokens.empty()) throw std::runtime_error("a program must contain at least one instruction");
        if (tokens[0] != "begin") throw std::runtime_error("a program must start with a 'begin' instruction");
        if (tokens.back() != "end") throw std::runtime_error("a program must end with an 'end' instruction");
        std::vector<Block> root_blocks;
        size_t i = parse_branch(root_blocks, 0);
        if (i < tokens.size() - 1) throw std::runtime_error("dangling instructions after program end at " + std::to_string(i));
        return Program::from_root_blocks(root_blocks);
    }
};

// ---- program inputs (programs/inputs.rs) ----------------------------------------------------------------------------------
struct ProgramInputs {
    vec pub, secret_a, secret_b;
    ProgramInputs() {}
    ProgramInputs(const vec& p, const vec& a, const vec& b) : pub(p), secret_a(a), secret_b(b) {     // inputs.rs:12
        if (pub.size() > MAX_PUBLIC_INPUTS) throw std::runtime_error("expected no more than 8 public inputs, but received " + std::to_string(pub.size()));
        if (a.size() < b.size()) throw std::runtime_error("number of primary secret inputs cannot be smaller than the number of secondary secret inputs");
    }
};

// ---- processor: decoder (processor/decoder/mod.rs) --------------------------------------------------------------
struct DecoderVM {
    size_t step = 0;
    vec op_counter;
    vec sponge_trace[4];
    u128 sponge[4] = {0, 0, 0, 0};
    vec cf_bits[3], ld_bits[5], hd_bits[2];
    std::vector<vec> ctx_stack, loop_stack;
    size_t ctx_depth, loop_depth;

    explicit DecoderVM(size_t init_len) {                                // decoder/mod.rs:39
        op_counter.assign(init_len, 0);
        for (auto& r : sponge_trace) r.assign(init_len, 0);
        for (auto& r : cf_bits) r.assign(init_len, 0);
        for (auto& r : ld_bits) r.assign(init_len, 0);
        for (auto& r : hd_bits) r.assign(init_len, 0);
        ctx_stack.push_back(vec(init_len, 0));
        ctx_depth = ctx_stack.size();
        loop_depth = 0;
    }
    size_t trace_length() const { return op_counter.size(); }
    std::string at() const { return std::to_string(step); }
    template <class F> void for_all(F f) {
        f(op_counter);
        for (auto& r : sponge_trace) f(r);
        for (auto& r : cf_bits) f(r);
        for (auto& r : ld_bits) f(r);
        for (auto& r : hd_bits) f(r);
        for (auto& r : ctx_stack) f(r);
        for (auto& r : loop_stack) f(r);
    }
    void advance_step(bool is_user_op) {                                 // decoder/mod.rs:276
        step += 1;
        if (step >= trace_length()) { size_t nl = trace_length() * 2; for_all([&](vec& r) { r.resize(nl, 0); }); }
        op_counter[step] = is_user_op ? op_counter[step - 1] + 1 : op_counter[step - 1];
    }
    void set_op_bits(uint8_t flow, uint8_t user) {                       // decoder/mod.rs:303 (written at step - 1)
        size_t s = step - 1;
        for (int i = 0; i < 3; i++) cf_bits[i][s] = (flow >> i) & 1;
        for (int i = 0; i < 5; i++) ld_bits[i][s] = (user >> i) & 1;
        for (int i = 0; i < 2; i++) hd_bits[i][s] = (user >> (i + 5)) & 1;
    }
    void save_context() {                                                // decoder/mod.rs:327
        ctx_depth += 1;
        if (ctx_depth > MAX_CONTEXT_DEPTH) throw std::runtime_error("context stack overflow at step " + at());
        if (ctx_depth > ctx_stack.size()) ctx_stack.push_back(vec(trace_length(), 0));
        for (size_t i = 1; i < ctx_stack.size(); i++) ctx_stack[i][step] = ctx_stack[i - 1][step - 1];
        ctx_stack[0][step] = sponge[0];
    }
    u128 pop_context() {                                                 // decoder/mod.rs:350
        if (ctx_depth == 0) throw std::runtime_error("context stack underflow at step " + at());
        for (size_t i = 1; i < ctx_stack.size(); i++) ctx_stack[i - 1][step] = ctx_stack[i][step - 1];
        ctx_depth -= 1;
        return ctx_stack[0][step - 1];
    }
    void copy_context_stack() { for (auto& r : ctx_stack) r[step] = r[step - 1]; }   // decoder/mod.rs:366
    void save_loop_image(u128 image) {                                   // decoder/mod.rs:375
        loop_depth += 1;
        if (loop_depth > MAX_LOOP_DEPTH) throw std::runtime_error("loop stack overflow at step " + at());
        if (loop_depth > loop_stack.size()) loop_stack.push_back(vec(trace_length(), 0));
        for (size_t i = 1; i < loop_stack.size(); i++) loop_stack[i][step] = loop_stack[i - 1][step - 1];
        loop_stack[0][step] = image;
    }
    u128 peek_loop_image() {                                             // decoder/mod.rs:397
        if (loop_depth == 0) throw std::runtime_error("loop stack underflow at step " + at());
        for (auto& r : loop_stack) r[step] = r[step - 1];
        return loop_stack[0][step];
    }
    u128 pop_loop_image() {                                              // decoder/mod.rs:411
        if (loop_depth == 0) throw std::runtime_error("loop stack underflow at step " + at());
        for (size_t i = 1; i < loop_stack.size(); i++) loop_stack[i - 1][step] = loop_stack[i][step - 1];
        loop_depth -= 1;
        return loop_stack[0][step - 1];
    }
    void copy_loop_stack() { for (auto& r : loop_stack) r[step] = r[step - 1]; }     // decoder/mod.rs:427
    void set_sponge(u128 a, u128 b, u128 c, u128 d) {                    // decoder/mod.rs:438
        sponge[0] = a; sponge[1] = b; sponge[2] = c; sponge[3] = d;
        for (int i = 0; i < 4; i++) sponge_trace[i][step] = sponge[i];
    }
    void start_block() {                                                 // decoder/mod.rs:160
        if (step % BASE_CYCLE_LENGTH != BASE_CYCLE_LENGTH - 1) throw std::runtime_error("cannot start context block at step " + at() + ": operation alignment is not valid");
        advance_step(false); save_context(); copy_loop_stack();
        set_op_bits(F_BEGIN, OP_NOOP);
        set_sponge(0, 0, 0, 0);
    }
    void end_block(u128 sibling_hash, bool true_branch) {                // decoder/mod.rs:172
        if (step % BASE_CYCLE_LENGTH != 0) throw std::runtime_error("cannot exit context block at step " + at() + ": operation alignment is not valid");
        advance_step(false);
        u128 context_hash = pop_context();
        copy_loop_stack();
        u128 block_hash = sponge[0];
        if (true_branch) { set_op_bits(F_TEND, OP_NOOP); set_sponge(context_hash, block_hash, sibling_hash, 0); }
        else { set_op_bits(F_FEND, OP_NOOP); set_sponge(context_hash, sibling_hash, block_hash, 0); }
    }
    void start_loop(u128 image) {                                        // decoder/mod.rs:194
        if (step % BASE_CYCLE_LENGTH != BASE_CYCLE_LENGTH - 1) throw std::runtime_error("cannot start a loop at step " + at() + ": operation alignment is not valid");
        advance_step(false); save_context(); save_loop_image(image);
        set_op_bits(F_LOOP, OP_NOOP);
        set_sponge(0, 0, 0, 0);
    }
    void wrap_loop() {                                                   // decoder/mod.rs:206
        if (step % BASE_CYCLE_LENGTH != BASE_CYCLE_LENGTH - 1) throw std::runtime_error("cannot wrap a loop at step " + at() + ": operation alignment is not valid");
        advance_step(false); copy_context_stack();
        if (sponge[0] != peek_loop_image()) throw std::runtime_error("cannot wrap a loop at step " + at() + ": hash of the last iteration doesn't match loop image");
        set_op_bits(F_WRAP, OP_NOOP);
        set_sponge(0, 0, 0, 0);
    }
    void break_loop() {                                                  // decoder/mod.rs:219
        if (step % BASE_CYCLE_LENGTH != BASE_CYCLE_LENGTH - 1) throw std::runtime_error("cannot break a loop at step " + at() + ": operation alignment is not valid");
        advance_step(false); copy_context_stack();
        if (sponge[0] != pop_loop_image()) throw std::runtime_error("cannot break a loop at step " + at() + ": hash of the last iteration doesn't match loop image");
        set_op_bits(F_BREAK, OP_NOOP);
        set_sponge(sponge[0], sponge[1], sponge[2], sponge[3]);
    }
    void decode_op(UserOp op, u128 op_value) {                           // decoder/mod.rs:232
        if (op_value != 0) {
            if (op != OP_PUSH) throw std::runtime_error(std::string("invalid ") + op_name(op) + " operation at step " + at() + ": op_value is non-zero");
            if (step % 8 != 0) throw std::runtime_error("invalid PUSH operation alignment at step " + at());
        }
        advance_step(true); copy_context_stack(); copy_loop_stack();
        set_op_bits(F_HACC, op);
        sponge_apply_round(sponge, (u128)op, op_value, step - 1);        // decoder/mod.rs:448
        for (int i = 0; i < 4; i++) sponge_trace[i][step] = sponge[i];
    }
    static void fill_register(vec& r, size_t from, u128 value) { size_t to = r.size(); r.resize(from, 0); r.resize(to, value); }  // decoder/mod.rs:462
    void finalize_trace() {                                              // decoder/mod.rs:253
        u128 last = op_counter[step];
        fill_register(op_counter, step + 1, last);
        for (auto& r : cf_bits) fill_register(r, step, 1);
        for (auto& r : ld_bits) fill_register(r, step, 1);
        for (auto& r : hd_bits) fill_register(r, step, 1);
        for (auto& r : sponge_trace) { u128 v = r[step]; fill_register(r, step + 1, v); }
        for (auto& r : ctx_stack) { u128 v = r[step]; fill_register(r, step + 1, v); }
        for (auto& r : loop_stack) { u128 v = r[step]; fill_register(r, step + 1, v); }
        step = trace_length() - 1;
    }
    size_t max_ctx_stack_depth() const { return ctx_stack.size() - 1; }  // decoder/mod.rs:93
    size_t max_loop_stack_depth() const { return loop_stack.size(); }    // decoder/mod.rs:99
    std::vector<vec> into_register_traces() {                            // decoder/mod.rs:120
        std::vector<vec> regs;
        regs.push_back(op_counter);
        for (auto& r : sponge_trace) regs.push_back(r);
        for (auto& r : cf_bits) regs.push_back(r);
        for (auto& r : ld_bits) regs.push_back(r);
        for (auto& r : hd_bits) regs.push_back(r);
        ctx_stack.pop_back();
        for (auto& r : ctx_stack) regs.push_back(r);
        for (auto& r : loop_stack) regs.push_back(r);
        return regs;
    }
};

// ---- processor: user stack (processor/stack/mod.rs) -----------------------------------------------------------------
static inline bool is_pow2(u128 v) { return v != 0 && (v & (v - 1)) == 0; }
struct StackVM {
    std::vector<vec> registers;
    vec tape_a, tape_b;
    size_t max_depth, depth, step = 0;
    StackVM(const ProgramInputs& in, size_t init_len) {                  // stack/mod.rs:29
        size_t d = std::max(in.pub.size(), MIN_STACK_DEPTH);
        for (size_t i = 0; i < d; i++) { vec r(init_len, 0); if (i < in.pub.size()) r[0] = in.pub[i]; registers.push_back(r); }
        tape_a.assign(in.secret_a.rbegin(), in.secret_a.rend());        // reversed: consumed in FIFO order by pop()
        tape_b.assign(in.secret_b.rbegin(), in.secret_b.rend());
        max_depth = depth = in.pub.size();
    }
    size_t trace_length() const { return registers[0].size(); }
    std::string at() const { return std::to_string(step); }
    u128 get_stack_top() const { return registers[0][step]; }            // stack/mod.rs:126
    void advance_step() { step += 1; if (step >= trace_length()) { size_t nl = trace_length() * 2; for (auto& r : registers) r.resize(nl, 0); } }  // stack/mod.rs:655
    void copy_state(size_t start) { for (size_t i = start; i < depth; i++) registers[i][step] = registers[i][step - 1]; }                        // stack/mod.rs:608
    void shift_left(size_t start, size_t cnt) {                          // stack/mod.rs:614
        if (depth < cnt) throw std::runtime_error("stack underflow at step " + at());
        for (size_t i = start; i < depth; i++) registers[i - cnt][step] = registers[i][step - 1];
        for (size_t i = depth - cnt; i < depth; i++) registers[i][step] = 0;
        depth -= cnt;
    }
    void shift_right(size_t start, size_t cnt) {                         // stack/mod.rs:631
        depth += cnt;
        if (depth > MAX_STACK_DEPTH) throw std::runtime_error("stack overflow at step " + at());
        if (depth > max_depth) {
            max_depth += cnt;
            if (max_depth > registers.size()) { size_t add_n = max_depth - registers.size(); for (size_t i = 0; i < add_n; i++) registers.push_back(vec(trace_length(), 0)); }
        }
        for (size_t i = start; i < depth - cnt; i++) registers[i + cnt][step] = registers[i][step - 1];
    }
    void need(size_t d) const { if (depth < d) throw std::runtime_error("stack underflow at step " + at()); }
    u128& cur(size_t i) { return registers[i][step]; }
    u128 old(size_t i) const { return registers[i][step - 1]; }
    static u128 pop(vec& t) { u128 v = t.back(); t.pop_back(); return v; }
    void bad_hint(const char* op) const { throw std::runtime_error(std::string("execution hint is not valid for ") + op + " operation"); }
    void need_binary_input(u128 bit) const { if (bit > 1) throw std::runtime_error("expected binary input at step " + at() + " but received: " + u128_to_string(bit)); }

    void execute(UserOp op, const OpHint& hint) {                        // stack/mod.rs:61
        advance_step();
        switch (op) {
            case OP_BEGIN: case OP_NOOP: copy_state(0); break;                                          // :151
            case OP_ASSERT:                                                                             // :155
                need(1);
                if (old(0) != 1) throw std::runtime_error("ASSERT failed at step " + at());
                shift_left(1, 1); break;
            case OP_ASSERTEQ:                                                                           // :162
                need(2);
                if (old(0) != old(1)) throw std::runtime_error("ASSERTEQ failed at step " + at());
                shift_left(2, 2); break;
            case OP_PUSH:                                                                               // :172
                shift_right(0, 1);
                if (hint.kind != H_PUSH_VALUE) throw std::runtime_error("invalid value for PUSH operation at step " + at());
                cur(0) = hint.v; break;
            case OP_READ:                                                                               // :181
                if (hint.kind == H_EQ_START) {
                    need(2);
                    u128 x = old(0), y = old(1);
                    tape_a.push_back(x == y ? (u128)1 : inv(sub(x, y)));
                } else if (hint.kind == H_NONE) {
                    if (tape_a.empty()) throw std::runtime_error("attempt to read from empty tape A at step " + at());
                } else bad_hint("READ");
                shift_right(0, 1);
                cur(0) = pop(tape_a); break;
            case OP_READ2:                                                                              // :209
                if (hint.kind == H_PMPATH_START) {
                    need(3);
                    size_t n = (size_t)hint.v - 1;
                    if (tape_a.size() < n) throw std::runtime_error("too few items on tape A for pmpath macro");
                    if (tape_b.size() < n) throw std::runtime_error("too few items on tape B for pmpath macro");
                    u128 idx = old(2);
                    vec v_a(tape_a.end() - n, tape_a.end());
                    tape_a.resize(tape_a.size() - n);
                    for (size_t i = 0; i < n; i++) { tape_a.push_back((idx >> (n - i - 1)) & 1); tape_a.push_back(v_a[i]); }
                } else if (hint.kind == H_NONE) {
                    if (tape_a.empty()) throw std::runtime_error("attempt to read from empty tape A at step " + at());
                    if (tape_b.empty()) throw std::runtime_error("attempt to read from empty tape B at step " + at());
                } else bad_hint("READ2");
                shift_right(0, 2);
                { u128 va = pop(tape_a), vb = pop(tape_b); cur(0) = vb; cur(1) = va; }
                break;
            case OP_DUP: need(1); shift_right(0, 1); cur(0) = old(0); break;                            // :249
            case OP_DUP2: need(2); shift_right(0, 2); cur(0) = old(0); cur(1) = old(1); break;          // :255
            case OP_DUP4: need(4); shift_right(0, 4); for (int i = 0; i < 4; i++) cur(i) = old(i); break;   // :262
            case OP_PAD2: shift_right(0, 2); cur(0) = 0; cur(1) = 0; break;                             // :271
            case OP_DROP: need(1); shift_left(1, 1); break;                                             // :277
            case OP_DROP4: need(4); shift_left(4, 4); break;                                            // :282
            case OP_SWAP: need(2); cur(0) = old(1); cur(1) = old(0); copy_state(2); break;              // :287
            case OP_SWAP2: need(4); cur(0) = old(2); cur(1) = old(3); cur(2) = old(0); cur(3) = old(1); copy_state(4); break;   // :294
            case OP_SWAP4: need(8); for (int i = 0; i < 4; i++) { cur(i) = old(4 + i); cur(4 + i) = old(i); } copy_state(8); break;  // :303
            case OP_ROLL4: need(4); cur(0) = old(3); for (int i = 1; i < 4; i++) cur(i) = old(i - 1); copy_state(4); break;    // :316
            case OP_ROLL8: need(8); cur(0) = old(7); for (int i = 1; i < 8; i++) cur(i) = old(i - 1); copy_state(8); break;    // :325
            case OP_CHOOSE: {                                                                           // :340
                need(3);
                u128 c = old(2);
                if (c == 1) cur(0) = old(0); else if (c == 0) cur(0) = old(1);
                else throw std::runtime_error("CHOOSE on a non-binary condition at step " + at());
                shift_left(3, 2); break;
            }
            case OP_CHOOSE2: {                                                                          // :355
                need(6);
                u128 c = old(4);
                if (c == 1) { cur(0) = old(0); cur(1) = old(1); } else if (c == 0) { cur(0) = old(2); cur(1) = old(3); }
                else throw std::runtime_error("CHOOSE2 on a non-binary condition at step " + at());
                shift_left(6, 4); break;
            }
            case OP_CSWAP2: {                                                                           // :372
                need(6);
                u128 c = old(4);
                if (c == 0) { for (int i = 0; i < 4; i++) cur(i) = old(i); }
                else if (c == 1) { cur(0) = old(2); cur(1) = old(3); cur(2) = old(0); cur(3) = old(1); }
                else throw std::runtime_error("CSWAP2 on a non-binary condition at step " + at());
                shift_left(6, 2); break;
            }
            case OP_ADD: { need(2); cur(0) = add(old(0), old(1)); shift_left(2, 1); break; }            // :395
            case OP_MUL: { need(2); cur(0) = mul(old(0), old(1)); shift_left(2, 1); break; }            // :403
            case OP_INV: {                                                                              // :411
                need(1);
                if (old(0) == 0) throw std::runtime_error("cannot compute INV of 0 at step " + at());
                cur(0) = inv(old(0)); copy_state(1); break;
            }
            case OP_NEG: need(1); cur(0) = neg(old(0)); copy_state(1); break;                           // :419
            case OP_NOT: {                                                                              // :426
                need(1);
                if (old(0) > 1) throw std::runtime_error("cannot compute NOT of a non-binary value at step " + at());
                cur(0) = sub(1, old(0)); copy_state(1); break;
            }
            case OP_AND: {                                                                              // :434
                need(2);
                u128 x = old(0), y = old(1);
                if (x > 1 || y > 1) throw std::runtime_error("cannot compute AND for a non-binary value at step " + at());
                cur(0) = (x == 1 && y == 1) ? 1 : 0; shift_left(2, 1); break;
            }
            case OP_OR: {                                                                               // :445
                need(2);
                u128 x = old(0), y = old(1);
                if (x > 1 || y > 1) throw std::runtime_error("cannot compute OR for a non-binary value at step " + at());
                cur(0) = (x == 1 || y == 1) ? 1 : 0; shift_left(2, 1); break;
            }
            case OP_EQ: {                                                                               // :459
                need(3);
                u128 aux = old(0), x = old(1), y = old(2);
                if (x == y) cur(0) = 1;
                else {
                    if (aux != inv(sub(x, y))) throw std::runtime_error("invalid AUX value for EQ operation at step " + at());
                    cur(0) = 0;
                }
                shift_left(3, 2); break;
            }
            case OP_CMP: {                                                                              // :474
                if (hint.kind == H_CMP_START) {
                    need(10);
                    u128 a_val = old(8), b_val = old(9);
                    for (uint32_t i = 0; i < (uint32_t)hint.v; i++) { tape_a.push_back((a_val >> i) & 1); tape_b.push_back((b_val >> i) & 1); }
                } else if (hint.kind == H_NONE) {
                    need(8);
                    if (tape_a.empty()) throw std::runtime_error("attempt to read from empty tape A at step " + at());
                    if (tape_b.empty()) throw std::runtime_error("attempt to read from empty tape B at step " + at());
                } else bad_hint("CMP");
                u128 a_bit = pop(tape_a); need_binary_input(a_bit);
                u128 b_bit = pop(tape_b); need_binary_input(b_bit);
                u128 bit_gt = mul(a_bit, sub(1, b_bit)), bit_lt = mul(b_bit, sub(1, a_bit));
                u128 pw = old(0);
                if (!is_pow2(pw)) throw std::runtime_error("expected top of the stack at step " + at() + " to be a power of 2, but received " + u128_to_string(pw));
                u128 next_pw = pw == 1 ? div(pw, 2) : pw >> 1;
                u128 gt = old(4), lt = old(5);
                u128 not_set = mul(sub(1, gt), sub(1, lt));
                cur(0) = next_pw; cur(1) = a_bit; cur(2) = b_bit; cur(3) = not_set;
                cur(4) = add(gt, mul(bit_gt, not_set)); cur(5) = add(lt, mul(bit_lt, not_set));
                cur(6) = add(old(6), mul(b_bit, pw)); cur(7) = add(old(7), mul(a_bit, pw));
                copy_state(8); break;
            }
            case OP_BINACC: {                                                                           // :537
                if (hint.kind == H_RC_START) {
                    need(5);
                    u128 val = old(4);
                    uint32_t n = (uint32_t)hint.v;
                    for (uint32_t i = 0; i < n; i++) tape_a.push_back((val >> (n - i - 1)) & 1);
                } else if (hint.kind == H_NONE) {
                    need(4);
                    if (tape_a.empty()) throw std::runtime_error("attempt to read from empty tape A at step " + at());
                } else bad_hint("BINACC");
                u128 bit = pop(tape_a); need_binary_input(bit);
                u128 pw = old(2);
                if (!is_pow2(pw)) throw std::runtime_error("expected 3rd value from the top of the stack at step " + at() + " to be a power of 2, but received " + u128_to_string(pw));
                u128 acc = old(3);
                cur(0) = bit; cur(1) = 0; cur(2) = mul(pw, 2); cur(3) = add(acc, mul(bit, pw));
                copy_state(4); break;
            }
            case OP_RESCR: {                                                                            // :582
                need(HASH_STATE_WIDTH);
                u128 st[6]; for (int i = 0; i < 6; i++) st[i] = old(i);
                hasher_apply_round(st, step - 1);
                for (int i = 0; i < 6; i++) cur(i) = st[i];
                copy_state(HASH_STATE_WIDTH); break;
            }
            default: throw std::runtime_error("oracle VM: unknown operation");
        }
    }
    void finalize_trace() {                                              // stack/mod.rs:132
        size_t tl = trace_length();
        for (auto& r : registers) { r.resize(step + 1, 0); u128 v = r[step]; r.resize(tl, v); }
        step = tl - 1;
    }
    std::vector<vec> into_register_traces() { registers.resize(max_depth); return registers; }   // stack/mod.rs:144
};

// Rescue digest of up to 4 elements                                                                   utils/hasher.rs:12-26
static inline std::array<u128, 2> hasher_digest(const vec& values) {
    if (values.size() > 4) throw std::runtime_error("expected no more than 4, but received " + std::to_string(values.size()));
    u128 st[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < values.size(); i++) st[i] = values[i];
    std::reverse(st, st + 6);
    for (size_t i = 0; i < 10; i++) hasher_apply_round(st, i);
    std::reverse(st, st + 6);
    return {st[0], st[1]};
}

// ---- processor::execute (processor/mod.rs:23-179) ---------------------------------------------------------------------
struct ExecutionTrace { std::vector<vec> registers; size_t ctx_depth, loop_depth; };

static inline void vm_noop(DecoderVM& d, StackVM& s) { d.decode_op(OP_NOOP, 0); s.execute(OP_NOOP, OpHint()); }
static inline void vm_start_block(DecoderVM& d, StackVM& s) { d.start_block(); s.execute(OP_NOOP, OpHint()); }      // processor/mod.rs:118
static inline void vm_close_block(DecoderVM& d, StackVM& s, u128 sibling_hash, bool true_branch) {    // processor/mod.rs:125
    vm_noop(d, s);
    d.end_block(sibling_hash, true_branch); s.execute(OP_NOOP, OpHint());
    for (size_t i = 0; i < HACC_NUM_ROUNDS; i++) vm_noop(d, s);
}
static inline void vm_execute_span(const Span& b, DecoderVM& d, StackVM& s, bool is_first) {         // processor/mod.rs:99
    if (!is_first) vm_noop(d, s);
    for (size_t i = 0; i < b.ops.size(); i++) {
        OpHint h = b.get_hint(i);
        d.decode_op(b.ops[i], h.value());
        s.execute(b.ops[i], h);
    }
}
static void vm_execute_blocks(const std::vector<Block>& blocks, DecoderVM& d, StackVM& s);
static void vm_execute_loop(const Block& l, DecoderVM& d, StackVM& s) {                              // processor/mod.rs:146
    d.start_loop(loop_image(l)); s.execute(OP_NOOP, OpHint());
    for (;;) {
        vm_execute_blocks(l.body, d, s);
        u128 c = s.get_stack_top();
        if (c == 0) { d.break_loop(); s.execute(OP_NOOP, OpHint()); break; }
        else if (c == 1) { d.wrap_loop(); s.execute(OP_NOOP, OpHint()); }
        else throw std::runtime_error("cannot exit loop based on a non-binary condition " + u128_to_string(c));
    }
    if (!l.alt[0].is_span()) throw std::runtime_error("invalid skip block content: content must be a Span block");
    vm_execute_span(l.alt[0].span, d, s, true);
    vm_close_block(d, s, loop_skip_hash(l), true);
}
static void vm_execute_blocks(const std::vector<Block>& blocks, DecoderVM& d, StackVM& s) {           // processor/mod.rs:50
    if (!blocks[0].is_span()) throw std::runtime_error("first block in a sequence must be a Span block");
    vm_execute_span(blocks[0].span, d, s, true);
    for (size_t i = 1; i < blocks.size(); i++) {
        const Block& b = blocks[i];
        switch (b.kind) {
            case B_SPAN: vm_execute_span(b.span, d, s, false); break;
            case B_GROUP:
                vm_start_block(d, s);
                vm_execute_blocks(b.body, d, s);
                vm_close_block(d, s, 0, true);
                break;
            case B_SWITCH: {
                vm_start_block(d, s);
                u128 c = s.get_stack_top();
                if (c == 0) { vm_execute_blocks(b.alt, d, s); vm_close_block(d, s, seq_hash(b.body), false); }
                else if (c == 1) { vm_execute_blocks(b.body, d, s); vm_close_block(d, s, seq_hash(b.alt), true); }
                else throw std::runtime_error("cannot select a branch based on a non-binary condition " + u128_to_string(c));
                break;
            }
            case B_LOOP: {
                u128 c = s.get_stack_top();
                if (c == 0) { vm_start_block(d, s); vm_execute_blocks(b.alt, d, s); vm_close_block(d, s, loop_body_hash(b), false); }
                else if (c == 1) vm_execute_loop(b, d, s);
                else throw std::runtime_error("cannot enter loop based on a non-binary condition " + u128_to_string(c));
                break;
            }
        }
    }
}
static inline ExecutionTrace vm_execute(const Program& p, const ProgramInputs& inputs) {              // processor/mod.rs:23
    DecoderVM d(MIN_TRACE_LENGTH);
    StackVM s(inputs, MIN_TRACE_LENGTH);
    vm_execute_blocks(p.root.body, d, s);
    vm_close_block(d, s, 0, true);
    d.finalize_trace();
    s.finalize_trace();
    ExecutionTrace t;
    t.ctx_depth = d.max_ctx_stack_depth();
    t.loop_depth = d.max_loop_stack_depth();
    t.registers = d.into_register_traces();
    auto sr = s.into_register_traces();
    t.registers.insert(t.registers.end(), sr.begin(), sr.end());
    return t;
}
static inline ExecutionTrace vm_execute(const Program& p, const vec& public_inputs) { return vm_execute(p, ProgramInputs(public_inputs, {}, {})); }

// the Fibonacci example program (examples/fibonacci.rs:32-47): n-th term, inputs [1, 0], 1 output
static inline std::string fibonacci_source(size_t n_terms) {
    return "begin repeat." + std::to_string(n_terms - 1) + " swap dup.2 drop add end end";
}

}  // namespace orc
