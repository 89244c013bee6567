"""Makes tests/corpus_avi/: a valid AVI written by the library (tests/asan_host_harness write) and damaged copies of it - the cases a
reader of untrusted captures must survive: truncated RIFF sizes, `movi` without `idx1`, index entries past the end
of the file, zero-length chunks, 2^31-sized chunks, a stream header outside any stream list, the most negative height.
Deterministic; the files are committed (each below 2 KB).  usage: python tests/make_avi_corpus.py <harness executable>"""
import os
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "corpus_avi")


def find(data, tag, start=0):
    i = data.index(tag, start)
    return i, struct.unpack("<I", data[i + 4:i + 8])[0]


def put32(data, pos, v):
    return data[:pos] + struct.pack("<I", v & 0xFFFFFFFF) + data[pos + 4:]


def variants(base: bytes):
    out = {"valid": base}
    idx, idx_size = find(base, b"idx1")
    movi = base.index(b"movi")
    movi_list = movi - 8
    first = base.index(b"00dc", movi)
    strf, _ = find(base, b"strf")
    strh, _ = find(base, b"strh")
    avih, _ = find(base, b"avih")
    out["truncated_half"] = base[:len(base) // 2]
    out["truncated_in_header"] = base[:avih + 20]
    out["truncated_12"] = base[:12]
    out["truncated_in_index"] = base[:idx + 8 + idx_size // 2]
    out["riff_size_small"] = put32(base, 4, 40)
    out["riff_size_huge"] = put32(base, 4, 0xFFFFFFF0)
    out["no_idx1"] = put32(base[:idx], 4, idx - 8)
    out["idx1_size_huge"] = put32(base, idx + 4, 0x7FFFFFF0)
    out["idx1_size_max"] = put32(base, idx + 4, 0xFFFFFFFF)
    out["idx1_size_odd"] = put32(base, idx + 4, idx_size - 7)
    b = base
    for k in range(0, idx_size // 16):  # every entry points past the end of the file
        b = put32(b, idx + 8 + 16 * k + 8, 0x7FFFFF00 + 16 * k)
    out["idx1_offsets_past_eof"] = b
    b = base
    for k in range(0, idx_size // 16):
        b = put32(b, idx + 8 + 16 * k + 12, 0x80000000)
    out["idx1_sizes_2g"] = b
    out["idx1_first_offset_wrong"] = put32(base, idx + 8 + 8, 3)
    out["idx1_offsets_from_file_start"] = b""  # filled below
    b = base
    for k in range(0, idx_size // 16):
        off = struct.unpack("<I", base[idx + 8 + 16 * k + 8:idx + 8 + 16 * k + 12])[0]
        b = put32(b, idx + 8 + 16 * k + 8, off + movi)
    out["idx1_offsets_from_file_start"] = b
    out["chunk_size_2g"] = put32(base, first + 4, 0x80000000)
    out["chunk_size_max"] = put32(base, first + 4, 0xFFFFFFFF)
    out["chunk_size_zero_all"] = base[:first] + b"".join(b"00dc" + struct.pack("<I", 0) for _ in range(40))
    out["movi_size_zero"] = put32(base, movi_list + 4, 4)
    out["movi_size_huge"] = put32(base, movi_list + 4, 0xFFFFFFF0)
    out["movi_size_huge_no_idx1"] = put32(out["no_idx1"], movi_list + 4, 0xFFFFFFF0)
    out["movi_garbage_no_idx1"] = out["no_idx1"][:first] + bytes((i * 37 + 11) & 255 for i in range(len(out["no_idx1"]) - first))
    out["strf_size_small"] = put32(base, strf + 4, 12)
    out["strf_height_int_min"] = put32(base, strf + 8 + 8, 0x80000000)
    out["strf_height_negative"] = put32(base, strf + 8 + 8, (-48) & 0xFFFFFFFF)
    out["strf_dims_max"] = put32(put32(base, strf + 8 + 4, 0xFFFFFFFF), strf + 8 + 8, 0x7FFFFFFF)
    out["strh_not_video"] = base[:strh + 8] + b"auds" + base[strh + 12:]
    out["strh_size_huge"] = put32(base, strh + 4, 0xFFFFFFF0)
    out["avih_size_small"] = put32(base, avih + 4, 8)
    hdrl = base.index(b"hdrl") - 8
    out["hdrl_size_huge"] = put32(base, hdrl + 4, 0xFFFFFFF0)
    strl = base.index(b"strl") - 8
    # the stream header and format directly in hdrl (no strl around them): the stream number stays at -1
    out["strh_outside_strl"] = base[:strl] + b"JUNK" + struct.pack("<I", 4) + b"strl" + base[strl + 12:]
    out["list_nesting"] = base[:12] + b"".join(b"LIST" + struct.pack("<I", 0x7FFFFFF0) + b"hdrl" for _ in range(60)) + base[12:]
    out["not_riff"] = b"RIFX" + base[4:]
    out["not_avi"] = base[:8] + b"WAVE" + base[12:]
    out["empty"] = b""
    out["all_ff"] = b"\xff" * 400
    return out


def main():
    harness = sys.argv[1]
    os.makedirs(OUT, exist_ok=True)
    valid = os.path.join(OUT, "valid.avi")
    subprocess.check_call([harness, "write", valid])
    base = open(valid, "rb").read()
    for name, data in variants(base).items():
        with open(os.path.join(OUT, name + ".avi"), "wb") as f:
            f.write(data)
    print(len(os.listdir(OUT)), "files in", OUT)


if __name__ == "__main__":
    main()
