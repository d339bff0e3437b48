"""Test infrastructure: the reference's message schema (gubernator.proto:137-203, peers.proto:36-49) declared
through descriptor_pb2 so that the python protobuf runtime — an implementation independent of
gubernator_amd/csrc/wire.cpp — can produce request payloads and parse / re-serialize response payloads.
Field names, numbers and types are transcribed from the reference .proto files; nothing is generated."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, proto3_optional=False, oneof_index=None):
    f = msg.field.add(name=name, number=number, type=ftype, label=label)
    if type_name:
        f.type_name = type_name
    if proto3_optional:
        f.proto3_optional = True
        f.oneof_index = oneof_index
    return f


def _map_entry(msg, name):
    e = msg.nested_type.add(name=name)
    e.options.map_entry = True
    _field(e, "key", 1, F.TYPE_STRING)
    _field(e, "value", 2, F.TYPE_STRING)


def build():
    fd = descriptor_pb2.FileDescriptorProto(name="gubernator_test_schema.proto", package="pb.gubernator", syntax="proto3")
    en = fd.enum_type.add(name="Algorithm")
    for n, v in (("TOKEN_BUCKET", 0), ("LEAKY_BUCKET", 1)):
        en.value.add(name=n, number=v)
    en = fd.enum_type.add(name="Behavior")
    for n, v in (("BATCHING", 0), ("NO_BATCHING", 1), ("GLOBAL", 2), ("DURATION_IS_GREGORIAN", 4), ("RESET_REMAINING", 8),
                 ("MULTI_REGION", 16), ("DRAIN_OVER_LIMIT", 32)):
        en.value.add(name=n, number=v)
    en = fd.enum_type.add(name="Status")
    for n, v in (("UNDER_LIMIT", 0), ("OVER_LIMIT", 1)):
        en.value.add(name=n, number=v)

    m = fd.message_type.add(name="RateLimitReq")                       # gubernator.proto:137-182
    _field(m, "name", 1, F.TYPE_STRING)
    _field(m, "unique_key", 2, F.TYPE_STRING)
    _field(m, "hits", 3, F.TYPE_INT64)
    _field(m, "limit", 4, F.TYPE_INT64)
    _field(m, "duration", 5, F.TYPE_INT64)
    _field(m, "algorithm", 6, F.TYPE_ENUM, type_name=".pb.gubernator.Algorithm")
    _field(m, "behavior", 7, F.TYPE_ENUM, type_name=".pb.gubernator.Behavior")
    _field(m, "burst", 8, F.TYPE_INT64)
    _map_entry(m, "MetadataEntry")
    _field(m, "metadata", 9, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.RateLimitReq.MetadataEntry")
    m.oneof_decl.add(name="_created_at")
    _field(m, "created_at", 10, F.TYPE_INT64, proto3_optional=True, oneof_index=0)

    m = fd.message_type.add(name="RateLimitResp")                      # gubernator.proto:189-203
    _field(m, "status", 1, F.TYPE_ENUM, type_name=".pb.gubernator.Status")
    _field(m, "limit", 2, F.TYPE_INT64)
    _field(m, "remaining", 3, F.TYPE_INT64)
    _field(m, "reset_time", 4, F.TYPE_INT64)
    _field(m, "error", 5, F.TYPE_STRING)
    _map_entry(m, "MetadataEntry")
    _field(m, "metadata", 6, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.RateLimitResp.MetadataEntry")

    m = fd.message_type.add(name="GetRateLimitsReq")                   # gubernator.proto:45-47
    _field(m, "requests", 1, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.RateLimitReq")
    m = fd.message_type.add(name="GetRateLimitsResp")                  # gubernator.proto:50-54
    _field(m, "responses", 1, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.RateLimitResp")
    m = fd.message_type.add(name="GetPeerRateLimitsReq")               # peers.proto:36-41
    _field(m, "requests", 1, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.RateLimitReq")
    m = fd.message_type.add(name="GetPeerRateLimitsResp")              # peers.proto:43-47
    _field(m, "rate_limits", 1, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.RateLimitResp")

    m = fd.message_type.add(name="UpdatePeerGlobal")                   # peers.proto:55-62
    _field(m, "key", 1, F.TYPE_STRING)
    _field(m, "status", 2, F.TYPE_MESSAGE, type_name=".pb.gubernator.RateLimitResp")
    _field(m, "algorithm", 3, F.TYPE_ENUM, type_name=".pb.gubernator.Algorithm")
    _field(m, "duration", 4, F.TYPE_INT64)
    _field(m, "created_at", 5, F.TYPE_INT64)
    m = fd.message_type.add(name="UpdatePeerGlobalsReq")               # peers.proto:51-53
    _field(m, "globals", 1, F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".pb.gubernator.UpdatePeerGlobal")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    names = ["RateLimitReq", "RateLimitResp", "GetRateLimitsReq", "GetRateLimitsResp", "GetPeerRateLimitsReq", "GetPeerRateLimitsResp",
             "UpdatePeerGlobal", "UpdatePeerGlobalsReq"]
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("pb.gubernator." + n)) for n in names}


PB = build()
